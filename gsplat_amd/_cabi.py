"""ctypes binding of the C-ABI in include/gsplat_amd.h (libgsplat_amd.so).

This is the only place that crosses from Python into native code. There is NO CPU fallback:
if the shared library is missing the import fails loudly, and every call targets HIP kernels
on the current device/stream.
"""
from __future__ import annotations

import ctypes
import json
import os
import threading
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_ENV = "GSPLAT_AMD_LIB"  # optional override of the library path (e.g. a debug build)


class GsplatAmdError(RuntimeError):
    pass


def lib_path() -> str:
    return os.environ.get(_LIB_ENV) or os.path.join(_HERE, "csrc", "libgsplat_amd.so")


def _load() -> ctypes.CDLL:
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            f"gsplat_amd: native library not found at {path}. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C gsplat_amd/csrc` "
            "(hipcc, --offload-arch=gfx950). There is no CPU fallback."
        )
    return ctypes.CDLL(path)


_lib = _load()

# signature table: p = pointer (device pointer or NULL), u = uint32, i = int, l = int64, f = float, d = double
_P, _U, _I, _L, _F = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int64, ctypes.c_float
_CODES = {"p": _P, "u": _U, "i": _I, "l": _L, "f": _F, "d": ctypes.c_double}

_HEADER = os.path.join(os.path.dirname(_HERE), "include", "gsplat_amd.h")


def _parse_header(path: str):
    """Derive the ctypes signatures from include/gsplat_amd.h so that the header stays the single
    source of truth. Returns {name: (restype_code, [arg codes])} for every `gsx_*` prototype."""
    import re

    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    sigs = {}
    for m in re.finditer(r"\b(int64_t|int|const char \*|void \*|void)\s*(gsx_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        codes = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    codes.append("p")
                elif a.startswith("uint32_t"):
                    codes.append("u")
                elif a.startswith("int64_t"):
                    codes.append("l")
                elif a.startswith("float"):
                    codes.append("f")
                elif a.startswith("double"):
                    codes.append("d")
                elif a.startswith("int "):
                    codes.append("i")
                else:
                    raise ImportError(f"gsplat_amd: cannot parse argument '{a}' of {name} in {path}")
        sigs[name] = ({"int": "i", "int64_t": "l", "void *": "p", "void": "v"}.get(ret, "s"), codes)
    return sigs


_SIG_TABLE = os.path.join(_HERE, "csrc", "abi_signatures.json")  # written by write_signature_table() at build time


def _load_signatures():
    """The header is the source of truth in a checkout; an installed copy of the package (no ../include) reads the table
    that build() generated from it next to the library."""
    if os.path.exists(_HEADER):
        return _parse_header(_HEADER)
    if os.path.exists(_SIG_TABLE):
        with open(_SIG_TABLE) as f:
            return {k: (v[0], list(v[1])) for k, v in json.load(f).items()}
    raise ImportError(f"gsplat_amd: neither {_HEADER} nor {_SIG_TABLE} found; rebuild with __graft_entry__.build()")


def write_signature_table() -> str:
    with open(_SIG_TABLE, "w") as f:
        json.dump({k: [v[0], v[1]] for k, v in _parse_header(_HEADER).items()}, f)
    return _SIG_TABLE


SIGNATURES = _load_signatures()


def _bind():
    for name, (ret, codes) in SIGNATURES.items():
        fn = getattr(_lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = {"i": ctypes.c_int, "l": _L, "s": ctypes.c_char_p, "p": ctypes.c_void_p, "v": None}[ret]
        fn.argtypes = [_CODES[c] for c in codes]


_bind()

ABI_VERSION = _lib.gsx_version()
ARCH = _lib.gsx_arch().decode()


def exported_symbols():
    return list(SIGNATURES)


_tls = threading.local()  # .dev = device index of the tensors marshalled for the next call (per thread: _bwd ops run on
#                           autograd worker threads)


def ptr(t: Optional[torch.Tensor]):
    """Device pointer of a tensor (None -> NULL). The tensor must be contiguous and on the GPU. Remembers the tensor's
    device so that call() launches on THAT device's current stream (the reference's DEVICE_GUARD, Common.h:40)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise GsplatAmdError(
            "gsplat_amd kernels only run on a ROCm device (got a CPU tensor); there is no CPU fallback"
        )
    if not t.is_contiguous():
        raise GsplatAmdError("gsplat_amd: tensor must be contiguous")
    _tls.dev = t.device.index
    return t.data_ptr()


def ptr_host(t: torch.Tensor):
    """Address of a PINNED host tensor, for the few scalar results a kernel writes straight into host memory (pinned
    memory is mapped into the device's address space: no device-to-host copy kernel is needed for 8 bytes)."""
    if t.is_cuda or not t.is_pinned():
        raise GsplatAmdError("gsplat_amd: expected a pinned host tensor")
    return t.data_ptr()


def ptr_strided(t: torch.Tensor):
    """Device pointer of a tensor whose layout the caller has already checked (row-strided views)."""
    if not t.is_cuda:
        raise GsplatAmdError("gsplat_amd kernels only run on a ROCm device (got a CPU tensor); there is no CPU fallback")
    _tls.dev = t.device.index
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream() -> int:
    """Raw hipStream_t of torch's current stream on the current device. The private fast accessor costs ~0.3 us;
    torch.cuda.current_stream() builds a Stream object through several Python layers (~9 us, measured), which adds
    up to ~0.1 ms per step over the ~12 C-ABI calls of a step."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


_profile = None  # list of (entry point, start event, end event) while profiling is on
_profile_only = None  # optional set of entry points to record (None = all)


_torch_lib = None


def _compiled_ops_lib():
    """libgsplat_amd_torch.so (csrc/torch_ops.cpp), when it is loaded: its op bodies call the C-ABI without passing through
    call() below, so they carry their own event-pair hooks."""
    global _torch_lib
    if _torch_lib is None:
        path = os.environ.get("GSPLAT_AMD_TORCH_LIB") or os.path.join(_HERE, "csrc", "libgsplat_amd_torch.so")
        _torch_lib = False
        if os.path.exists(path):
            try:
                lib = ctypes.CDLL(path)
                lib.gsx_torch_profile_begin.argtypes = [ctypes.c_char_p]
                lib.gsx_torch_profile_begin.restype = None
                lib.gsx_torch_profile_end.restype = ctypes.c_char_p
                _torch_lib = lib
            except (OSError, AttributeError):
                _torch_lib = False
    return _torch_lib or None


def profile_begin(only=None) -> None:
    """Start recording a HIP-event pair around gsx_* calls (events go on the stream the kernels are launched on:
    torch's current stream). `only` = iterable of entry-point names restricts the recording (bench.py times just the
    dominant kernels inside its timed region so that event bookkeeping does not perturb the step time)."""
    global _profile, _profile_only
    _profile = []
    _profile_only = None if only is None else frozenset(only)
    lib = _compiled_ops_lib()
    if lib is not None:
        lib.gsx_torch_profile_begin(" ".join(sorted(_profile_only or ())).encode())


def profile_end() -> dict:
    """Stop recording; returns {entry point: [ms per call, ...]} (synchronises the device)."""
    global _profile
    rec, _profile = _profile or [], None
    torch.cuda.synchronize()
    out = {}
    for name, a, b in rec:
        out.setdefault(name, []).append(a.elapsed_time(b))
    lib = _compiled_ops_lib()
    if lib is not None:  # the calls made by the compiled op bodies
        for line in lib.gsx_torch_profile_end().decode().splitlines():
            name, ms = line.split()
            out.setdefault(name, []).append(float(ms))
    return out


def call(name: str, *args) -> None:
    """Invoke a gsx_* entry point on the current stream of the device that owns the tensors marshalled by ptr() for it
    (device guard: if that is not the current device it becomes current for the duration of the launch); raise on a
    non-zero return code."""
    dev = getattr(_tls, "dev", None)
    if dev is not None and dev != torch.cuda.current_device():
        _tls.dev = None
        with torch.cuda.device(dev):
            return call(name, *args)
    fn = getattr(_lib, name)
    if _profile is not None and (_profile_only is None or name in _profile_only):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(*args, current_stream())
        b.record()
        _profile.append((name, a, b))
    else:
        rc = fn(*args, current_stream())
    if rc != 0:
        msg = _lib.gsx_last_error().decode(errors="replace")
        if rc == -1:
            raise ValueError(f"{name}: {msg}")
        raise GsplatAmdError(f"{name} failed (code {rc}): {msg}")


def scan_workspace_bytes(n: int) -> int:
    return int(_lib.gsx_scan_workspace_bytes(n))


def sort_workspace_bytes(n: int) -> int:
    return int(_lib.gsx_sort_pairs_workspace_bytes(n))


def tile_sort_supported(n_images: int, tile_w: int, tile_h: int) -> bool:
    return bool(_lib.gsx_isect_tile_sort_supported(n_images, tile_w, tile_h))


def tile_sort_workspace_bytes(n: int, n_images: int, tile_w: int, tile_h: int) -> int:
    return int(_lib.gsx_isect_tile_sort_workspace_bytes(n, n_images, tile_w, tile_h))


def isect_fused_supported(n_images: int, tile_w: int, tile_h: int, packed: bool) -> bool:
    return bool(_lib.gsx_isect_fused_supported(n_images, tile_w, tile_h, int(packed)))


def isect_fused_count_workspace_bytes(rows: int, n_images: int, tile_w: int, tile_h: int) -> int:
    return int(_lib.gsx_isect_fused_count_workspace_bytes(rows, n_images, tile_w, tile_h))


def isect_fused_emit_workspace_bytes(n: int, n_images: int, tile_w: int, tile_h: int) -> int:
    return int(_lib.gsx_isect_fused_emit_workspace_bytes(n, n_images, tile_w, tile_h))


def isect_binned_supported(rows: int, n_images: int, tile_w: int, tile_h: int, packed: bool) -> bool:
    """Pure query (asking changes nothing)."""
    return bool(_lib.gsx_isect_binned_supported(rows, n_images, tile_w, tile_h, int(packed)))


def isect_binned_should_try(rows: int, n_images: int, tile_w: int, tile_h: int, packed: bool) -> bool:
    """The decision of ONE intersection (counts a skipped call while a retry note is active): call once, keep the answer."""
    return bool(_lib.gsx_isect_binned_should_try(rows, n_images, tile_w, tile_h, int(packed)))


class IsectPathMemory:
    """The retry notes of the tile-owner-major intersection path, owned by ONE caller (include/gsplat_amd.h:
    gsx_isect_path_memory_*). Which intersection kernel runs depends on recent history (a clustered scene that was sent back is
    not tried again for 63 calls); `with memory:` makes that history this object's for the calls inside, on this thread - a
    trainer and a viewer, or two trainers, each hold their own. Without one, a thread uses a private default."""

    def __init__(self):
        self._h = _lib.gsx_isect_path_memory_create()
        if not self._h:
            raise MemoryError("gsx_isect_path_memory_create failed")
        self._prev = []

    def __enter__(self):
        self._prev.append(_lib.gsx_isect_path_memory_use(self._h))
        return self

    def __exit__(self, *exc):
        _lib.gsx_isect_path_memory_use(self._prev.pop())
        return False

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.gsx_isect_path_memory_destroy(h)


def isect_binned_count_workspace_bytes(rows: int, n_images: int, tile_w: int, tile_h: int) -> int:
    return int(_lib.gsx_isect_binned_count_workspace_bytes(rows, n_images, tile_w, tile_h))


def isect_binned_emit_workspace_bytes(n: int) -> int:
    return int(_lib.gsx_isect_binned_emit_workspace_bytes(n))


def copy_column_groups(groups, rows: int) -> None:
    """One launch of gsx_copy_column_groups. `groups`: list of (src_ptr, src_row_stride, dst_ptr, dst_row_stride, width)
    in 32-bit words (device pointers as ints)."""
    n = len(groups)
    src = (ctypes.c_void_p * n)(*[g[0] for g in groups])
    dst = (ctypes.c_void_p * n)(*[g[2] for g in groups])
    ss = (ctypes.c_uint32 * n)(*[g[1] for g in groups])
    ds = (ctypes.c_uint32 * n)(*[g[3] for g in groups])
    w = (ctypes.c_uint32 * n)(*[g[4] for g in groups])
    call("gsx_copy_column_groups", n, src, ss, dst, ds, w, rows)


def copy_column_groups_mapped(groups, seg_rows, seg_n, map_dst: bool) -> None:
    """gsx_copy_column_groups_mapped: `groups` as in copy_column_groups; seg_rows[k] = C_local * N_k, seg_n[k] = N_k."""
    n, m = len(groups), len(seg_rows)
    src = (ctypes.c_void_p * n)(*[g[0] for g in groups])
    dst = (ctypes.c_void_p * n)(*[g[2] for g in groups])
    ss = (ctypes.c_uint32 * n)(*[g[1] for g in groups])
    ds = (ctypes.c_uint32 * n)(*[g[3] for g in groups])
    w = (ctypes.c_uint32 * n)(*[g[4] for g in groups])
    sr = (ctypes.c_int64 * m)(*[int(v) for v in seg_rows])
    sn = (ctypes.c_int64 * m)(*[int(v) for v in seg_n])
    call("gsx_copy_column_groups_mapped", n, src, ss, dst, ds, w, m, sr, sn, int(bool(map_dst)))


def copy_message_columns(msg_ptr: int, msg_stride: int, rows: int, groups, to_message: bool, seg_rows=None, seg_n=None) -> None:
    """gsx_copy_message_columns. `groups`: list of (column, width, field_ptr, field_row_stride) in 32-bit words."""
    n = len(groups)
    cols = (ctypes.c_uint32 * n)(*[g[0] for g in groups])
    w = (ctypes.c_uint32 * n)(*[g[1] for g in groups])
    fields = (ctypes.c_void_p * n)(*[g[2] for g in groups])
    fs = (ctypes.c_uint32 * n)(*[g[3] for g in groups])
    m = 0 if seg_rows is None else len(seg_rows)
    sr = (ctypes.c_int64 * max(m, 1))(*([int(v) for v in seg_rows] if m else [0]))
    sn = (ctypes.c_int64 * max(m, 1))(*([int(v) for v in seg_n] if m else [0]))
    call("gsx_copy_message_columns", ctypes.c_void_p(msg_ptr), msg_stride, rows, n, cols, w, fields, fs, int(bool(to_message)), m,
         sr, sn)


def sort_pairs(keys, vals, keys_alt, vals_alt, n: int, end_bit: int, workspace) -> bool:
    """Returns True when the sorted data ended up in the alt buffers."""
    flag = ctypes.c_int(0)
    rc = _lib.gsx_sort_pairs(ptr(keys), ptr(vals), ptr(keys_alt), ptr(vals_alt), n, end_bit, ptr(workspace),
                             workspace.numel() * workspace.element_size(), ctypes.addressof(flag), current_stream())
    if rc != 0:
        raise GsplatAmdError(f"gsx_sort_pairs failed (code {rc}): {_lib.gsx_last_error().decode(errors='replace')}")
    return bool(flag.value)
