"""On-disk formats of a trained Gaussian model: standard 3DGS ``.ply``, the antimatter15
``.splat`` records and the SuperSplat *compressed* ``.ply`` — the byte layouts of ``gsplat/exporter.py`` (reference
``export_splats`` ``:588-666``, ``splat2ply_bytes`` ``:378-432``, ``splat2splat_bytes`` ``:533-585``,
``splat2ply_bytes_compressed`` ``:209-375``), plus a dependency-free reader for the standard ``.ply``
(``load_ply_to_splats`` ``:435-530``; the reference needs the third-party ``plyfile`` package for it).

Same function names, arguments and bytes as the reference; the work is organised differently:

* every format is assembled from whole-model tensor ops on the tensors' device (the reference walks the compressed
  format chunk by chunk in Python — ~4 k iterations of ~40 small ops for 1 M splats, and the ``.splat`` writer appends
  32-byte records one at a time) — so on a GPU the model is quantised where it lives and only the packed bytes
  (16 B + K*3 B per splat instead of 4*(14 + 3K) B) cross PCIe;
* quantisation follows the reference operation by operation (same torch ops in the same order per element), so on CPU
  tensors the output is byte-identical to the reference's (tests/test_exporter.py, golden vectors produced by importing
  the reference: tests/golden/exporter_ref.npz).
"""
from __future__ import annotations

import math
from typing import Dict, Literal, Optional

import numpy as np
import torch
from torch import Tensor

_SH_C0 = 0.28209479177387814
_CHUNK = 256


def sh2rgb(sh: Tensor) -> Tensor:
    """Band-0 SH coefficient -> colour (reference exporter.py:25-35)."""
    return sh * _SH_C0 + 0.5


# ---- Morton order of the centres (reference :38-100) ----------------------------------------------------------------
def _spread3(x: Tensor) -> Tensor:
    """10 low bits of x, two zero bits between neighbours."""
    x = x & 0x000003FF
    x = (x ^ (x << 16)) & 0xFF0000FF
    x = (x ^ (x << 8)) & 0x0300F00F
    x = (x ^ (x << 4)) & 0x030C30C3
    x = (x ^ (x << 2)) & 0x09249249
    return x


def encode_morton3_vec(x: Tensor, y: Tensor, z: Tensor) -> Tensor:
    return (_spread3(z) << 2) + (_spread3(y) << 1) + _spread3(x)


def sort_centers(centers: Tensor, indices: Tensor) -> Tensor:
    """``indices`` reordered along the 30-bit Morton curve of the centres' bounding box (the grid coordinate 1024 of
    the maximum wraps to 0 under the 10-bit mask, as in the reference)."""
    lo, hi = centers.min(dim=0).values, centers.max(dim=0).values
    span = hi - lo
    span[span == 0] = 1
    grid = ((centers - lo) / span * 1024).floor().to(torch.int32)
    code = encode_morton3_vec(grid[:, 0], grid[:, 1], grid[:, 2])
    return indices[torch.argsort(code).to(indices.device)]


# ---- fixed-point packing (reference :103-206) -----------------------------------------------------------------------
def pack_unorm(value: Tensor, bits: int) -> Tensor:
    t = (1 << bits) - 1
    return torch.clamp((value * t + 0.5).floor(), min=0, max=t).to(torch.int64)


def pack_111011(x: Tensor, y: Tensor, z: Tensor) -> Tensor:
    return (pack_unorm(x, 11) << 21) | (pack_unorm(y, 10) << 11) | pack_unorm(z, 11)


def pack_8888(x: Tensor, y: Tensor, z: Tensor, w: Tensor) -> Tensor:
    return (pack_unorm(x, 8) << 24) | (pack_unorm(y, 8) << 16) | (pack_unorm(z, 8) << 8) | pack_unorm(w, 8)


def pack_rotation(q: Tensor) -> Tensor:
    """"Smallest three": index of the largest |component| in the top 2 bits, the other three (sign-normalised so that
    the dropped one is positive) as 10-bit values of c / sqrt(2) + 0.5."""
    q = q / torch.linalg.norm(q, dim=-1, keepdim=True)
    largest = torch.argmax(torch.abs(q), dim=-1)
    rows = torch.arange(q.size(0), device=q.device)
    q = torch.where((q[rows, largest] < 0)[:, None], q * -1, q)
    others = torch.tensor([[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]], dtype=torch.long, device=q.device)[largest]
    packed = pack_unorm(q[rows[:, None], others] * (math.sqrt(2) * 0.5) + 0.5, 10)
    return (largest.to(torch.int64) << 30) | (packed[:, 0] << 20) | (packed[:, 1] << 10) | packed[:, 2]


def _le_bytes(t: Tensor, dtype) -> bytes:
    return t.detach().cpu().numpy().astype(np.dtype(dtype).newbyteorder("<")).tobytes()


# ---- standard .ply (reference :378-432) -----------------------------------------------------------------------------
def splat2ply_bytes(means: Tensor, scales: Tensor, quats: Tensor, opacities: Tensor, sh0: Tensor, shN: Tensor) -> bytes:
    """Binary little-endian PLY, one float32 vertex record per splat: x y z | f_dc_0..2 | f_rest_* | opacity | scale_* |
    rot_*  (sh0 [N, 3], shN [N, 3*(K-1)] channel-major, as prepared by export_splats)."""
    head = ["ply", "format binary_little_endian 1.0", f"element vertex {means.shape[0]}"]
    head += [f"property float {a}" for a in "xyz"]
    head += [f"property float f_dc_{j}" for j in range(sh0.shape[1])]
    head += [f"property float f_rest_{j}" for j in range(shN.shape[1])]
    head += ["property float opacity"]
    head += [f"property float scale_{j}" for j in range(scales.shape[1])]
    head += [f"property float rot_{j}" for j in range(quats.shape[1])]
    head += ["end_header", ""]
    rows = torch.cat([means, sh0, shN, opacities.unsqueeze(1), scales, quats], dim=1).to(torch.float32)
    return "\n".join(head).encode() + _le_bytes(rows, np.float32)


def _parse_ply_header(raw: bytes):
    end = raw.find(b"end_header\n")
    if not raw.startswith(b"ply") or end < 0:
        raise ValueError("not a PLY file")
    lines = raw[:end].decode("ascii", errors="replace").splitlines()
    fmt, elements = None, []
    for ln in lines[1:]:
        tok = ln.split()
        if not tok or tok[0] == "comment":
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            elements.append((tok[1], int(tok[2]), []))
        elif tok[0] == "property":
            if tok[1] == "list":
                raise ValueError("PLY list properties are not part of the 3DGS vertex layout")
            elements[-1][2].append((tok[2], tok[1]))
    return fmt, elements, end + len(b"end_header\n")


_PLY_TYPES = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1",
              "char": "i1", "int8": "i1", "ushort": "u2", "uint16": "u2", "short": "i2", "int16": "i2", "uint": "u4",
              "uint32": "u4", "int": "i4", "int32": "i4"}


def load_ply_to_splats(path: str) -> Dict[str, Tensor]:
    """Read a standard 3DGS ``.ply`` (inverse of splat2ply_bytes; INRIA convention: ``f_rest_*`` channel-major) into
    float32 CPU tensors: means [N,3], scales [N,3] (log), quats [N,4] (as stored), opacities [N] (logit), sh0 [N,1,3],
    shN [N,K-1,3]. Reference exporter.py:435-530; binary PLY files of either endianness, no third-party parser."""
    with open(str(path), "rb") as f:
        raw = f.read()
    fmt, elements, off = _parse_ply_header(raw)
    if fmt not in ("binary_little_endian", "binary_big_endian"):
        raise ValueError(f"unsupported PLY format '{fmt}' (binary expected)")
    order = "<" if fmt == "binary_little_endian" else ">"
    name, count, props = elements[0]  # the reference reads ply.elements[0] as the vertex element
    dt = np.dtype([(p, order + _PLY_TYPES[t]) for p, t in props])
    vertex = np.frombuffer(raw, dtype=dt, count=count, offset=off)
    names = [p for p, _ in props]

    def cols(prefix):
        return sorted((p for p in names if p.startswith(prefix)), key=lambda s: int(s.split("_")[-1]))

    def stack(keys):
        return np.stack([np.asarray(vertex[k], dtype=np.float32) for k in keys], axis=1)

    rest = cols("f_rest_")
    if rest:
        if len(rest) % 3 != 0:
            raise ValueError(f"f_rest property count ({len(rest)}) is not a multiple of 3 (RGB channels); cannot "
                             "reshape SH coefficients.")
        shN = stack(rest).reshape(count, 3, len(rest) // 3).swapaxes(1, 2)  # channel-major -> (basis, channel)
    else:
        shN = np.zeros((count, 0, 3), dtype=np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()  # noqa: E731
    return {"means": t(stack(["x", "y", "z"])), "scales": t(stack(cols("scale_"))), "quats": t(stack(cols("rot_"))),
            "opacities": t(np.asarray(vertex["opacity"], dtype=np.float32)),
            "sh0": t(stack(["f_dc_0", "f_dc_1", "f_dc_2"])[:, None, :]), "shN": t(shN)}


# ---- .splat (reference :533-585) ------------------------------------------------------------------------------------
def splat2splat_bytes(means: Tensor, scales: Tensor, quats: Tensor, opacities: Tensor, sh0: Tensor) -> bytes:
    """32-byte records in Morton order: position 3 x f32 | linear scale 3 x f32 | RGBA u8 | rotation 4 x u8."""
    if means.shape[0] == 0:
        return b""
    colors = torch.cat([sh2rgb(sh0), torch.sigmoid(opacities).unsqueeze(-1)], dim=1)
    colors = (colors * 255).clamp(0, 255).to(torch.uint8)
    rots = ((quats / torch.linalg.norm(quats, dim=1, keepdim=True)) * 128 + 128).clamp(0, 255).to(torch.uint8)
    order = sort_centers(means, torch.arange(means.shape[0]))
    rec = np.empty(means.shape[0], dtype=[("p", "<f4", 3), ("s", "<f4", 3), ("c", "u1", 4), ("r", "u1", 4)])
    rec["p"] = means[order].detach().cpu().numpy()
    rec["s"] = torch.exp(scales)[order].detach().cpu().numpy()
    rec["c"] = colors[order].cpu().numpy()
    rec["r"] = rots[order].cpu().numpy()
    return rec.tobytes()


# ---- SuperSplat compressed .ply (reference :209-375) ----------------------------------------------------------------
def _encode_chunks(means: Tensor, scales: Tensor, quats: Tensor, opacities: Tensor, colors: Tensor):
    """Quantise `n` chunks of `S` splats at once: inputs [n, S, *]; returns (bounds float [n, 18], words int64 [n, S, 4])."""
    n, S = means.shape[0], means.shape[1]
    lo_m, hi_m = means.min(dim=1).values, means.max(dim=1).values
    lo_s = torch.clamp(scales.min(dim=1).values, -20, 20)
    hi_s = torch.clamp(scales.max(dim=1).values, -20, 20)
    lo_c, hi_c = colors.min(dim=1).values, colors.max(dim=1).values
    bounds = torch.cat([lo_m, hi_m, lo_s, hi_s, lo_c, hi_c], dim=1)
    nm = (means - lo_m[:, None]) / (hi_m - lo_m)[:, None]
    ns = (scales - lo_s[:, None]) / (hi_s - lo_s)[:, None]
    nc = (colors - lo_c[:, None]) / (hi_c - lo_c)[:, None]
    alpha = 1 / (1 + torch.exp(-opacities))
    words = torch.stack([
        pack_111011(nm[..., 0], nm[..., 1], nm[..., 2]),
        pack_rotation(quats.reshape(n * S, 4)).reshape(n, S),
        pack_111011(ns[..., 0], ns[..., 1], ns[..., 2]),
        pack_8888(nc[..., 0], nc[..., 1], nc[..., 2], alpha),
    ], dim=-1)
    return bounds, words


def splat2ply_bytes_compressed(means: Tensor, scales: Tensor, quats: Tensor, opacities: Tensor, sh0: Tensor, shN: Tensor,
                               chunk_max_size: int = _CHUNK, opacity_threshold: float = 1 / 255) -> bytes:
    """SuperSplat compressed PLY: splats with sigmoid(opacity) <= threshold dropped, the rest in Morton order in chunks
    of ``chunk_max_size``; per chunk 18 float bounds (position, clamped log-scale, colour), per splat four uint32 words
    (11-10-11 position, smallest-three rotation, 11-10-11 scale, 8-8-8-8 colour + opacity, all relative to the chunk's
    bounds), per splat K*3 bytes of higher-order SH (trunc((c / 8 + 0.5) * 256) clamped to a byte)."""
    keep = torch.sigmoid(opacities) > opacity_threshold
    means, scales, quats, opacities, shN = means[keep], scales[keep], quats[keep], opacities[keep], shN[keep]
    colors = sh2rgb(sh0)[keep]
    n = means.shape[0]
    n_chunks = n // chunk_max_size + (n % chunk_max_size != 0)
    order = (sort_centers(means, torch.arange(n)) if n > 0 else torch.arange(0)).to(means.device)
    means, scales, quats, opacities, colors, shN = (t[order] for t in (means, scales, quats, opacities, colors, shN))

    chunk_props = ["min_x", "min_y", "min_z", "max_x", "max_y", "max_z", "min_scale_x", "min_scale_y", "min_scale_z",
                   "max_scale_x", "max_scale_y", "max_scale_z", "min_r", "min_g", "min_b", "max_r", "max_g", "max_b"]
    head = ["ply", "format binary_little_endian 1.0", f"element chunk {n_chunks}"]
    head += [f"property float {p}" for p in chunk_props]
    head += [f"element vertex {n}"]
    head += [f"property uint {p}" for p in ("packed_position", "packed_rotation", "packed_scale", "packed_color")]
    head += [f"element sh {n}"]
    head += [f"property uchar f_rest_{j}" for j in range(shN.shape[1])]
    head += ["end_header", ""]

    bounds, words = [], []
    full = (n // chunk_max_size) * chunk_max_size
    for lo, hi, size in ((0, full, chunk_max_size), (full, n, n - full)):  # all full chunks at once, then the tail
        if hi > lo:
            b, w = _encode_chunks(*(t[lo:hi].reshape((-1, size) + t.shape[1:])
                                    for t in (means, scales, quats, opacities, colors)))
            bounds.append(b.reshape(-1))
            words.append(w.reshape(-1))
    sh_q = torch.clamp(torch.trunc((shN / 8 + 0.5) * 256), 0, 255).to(torch.uint8)
    body = b""
    if n > 0:
        body = _le_bytes(torch.cat(bounds), np.float32) + _le_bytes(torch.cat(words), np.uint32) + _le_bytes(
            sh_q.reshape(-1), np.uint8)
    return "\n".join(head).encode() + body


# ---- entry point (reference :588-666) -------------------------------------------------------------------------------
def export_splats(means: Tensor, scales: Tensor, quats: Tensor, opacities: Tensor, sh0: Tensor, shN: Tensor,
                  format: Literal["ply", "splat", "ply_compressed"] = "ply", save_to: Optional[str] = None) -> bytes:
    """Serialise a model (means [N,3], log-scales [N,3], quats [N,4], logit-opacities [N], sh0 [N,1,3], shN [N,K,3]);
    splats with a NaN / Inf in any attribute are dropped first. Returns the bytes; also writes them to ``save_to``."""
    N = means.shape[0]
    assert means.shape == (N, 3), "Means must be of shape (N, 3)"
    assert scales.shape == (N, 3), "Scales must be of shape (N, 3)"
    assert quats.shape == (N, 4), "Quaternions must be of shape (N, 4)"
    assert opacities.shape == (N,), "Opacities must be of shape (N,)"
    assert sh0.shape == (N, 1, 3), "sh0 must be of shape (N, 1, 3)"
    assert shN.ndim == 3 and shN.shape[0] == N and shN.shape[2] == 3, f"shN must be of shape (N, K, 3), got {shN.shape}"
    sh0 = sh0.squeeze(1)
    shN = shN.permute(0, 2, 1).reshape(N, 3 * shN.shape[1])  # channel-major [N, 3*K]
    finite = torch.ones(N, dtype=torch.bool, device=means.device)
    for t in (means, scales, quats, opacities[:, None], sh0, shN):
        finite &= torch.isfinite(t).all(dim=1)
    means, scales, quats, opacities, sh0, shN = (t[finite] for t in (means, scales, quats, opacities, sh0, shN))
    if format == "ply":
        data = splat2ply_bytes(means, scales, quats, opacities, sh0, shN)
    elif format == "splat":
        data = splat2splat_bytes(means, scales, quats, opacities, sh0)
    elif format == "ply_compressed":
        data = splat2ply_bytes_compressed(means, scales, quats, opacities, sh0, shN)
    else:
        raise ValueError(f"Unsupported format: {format}")
    if save_to:
        with open(save_to, "wb") as f:
            f.write(data)
    return data
