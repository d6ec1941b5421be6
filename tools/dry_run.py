"""CPU dry run of the host side of the pipeline (no GPU in the build container).

The product path has no CPU fallback, so nothing can be COMPUTED here — but everything above the kernels can be
EXERCISED: with device pointers taken from CPU tensors and launch failures ("no ROCm-capable device") ignored, a call of
rasterization() runs its orchestration, the op implementations, the ctypes marshalling of every C-ABI call (argument
count / types against include/gsplat_amd.h) and the host-side checks of the entry points (GSX_REQUIRE) until something
needs a value that only a kernel could have produced. Outputs are garbage; errors are real. Used by
tests/test_host_dry_run.py for the empty-scene path, whose control flow does not depend on kernel results.

    python tools/dry_run.py            # empty 3DGS / 2DGS scenes, dense and packed, forward + backward
"""
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def install():
    """Patch gsplat_amd for the dry run (this process only). Returns the list that collects the C-ABI calls made."""
    import gsplat_amd  # noqa: F401
    from gsplat_amd import _cabi, _ops

    calls = []
    _cabi.ptr = lambda t: None if t is None else t.data_ptr()
    _cabi.ptr_strided = lambda t: t.data_ptr()
    _cabi.current_stream = lambda: 0
    real_call = _cabi.call

    def call(name, *args):
        calls.append(name)
        try:
            real_call(name, *args)
        except _cabi.GsplatAmdError as e:  # a launch / memset without a device; host-side check failures are ValueError
            if not any(w in str(e) for w in ("device", "memset", "launch")):
                raise

    _cabi.call = call
    for mod in (_ops,):
        mod.ptr, mod.ptr_strided, mod.call = _cabi.ptr, _cabi.ptr_strided, call
    cpu = torch.library.Library("gsplat", "IMPL", "CPU")
    for name in _ops.SCHEMAS:
        cpu.impl(name, _ops.impl(name))
    install.keep = cpu
    return calls


def empty_scene(fn: str, packed: bool):
    import gsplat_amd
    from _util import make_scene

    W, H, C = 96, 64, 2
    names = ("means", "quats", "scales", "opacities", "colors")
    sc, _, _ = make_scene(N=8, C=C, width=W, height=H, seed=2)
    leaves = {k: sc[k][:0].clone().requires_grad_(True) for k in names}
    bg = torch.rand(C, 3)
    if fn == "3dgs":
        rc, ra, meta = gsplat_amd.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"],
                                                leaves["colors"], sc["viewmats"], sc["Ks"], W, H, packed=packed,
                                                backgrounds=bg, render_mode="RGB+ED", absgrad=True)
        (rc.sum() + ra.sum()).backward()
        assert rc.shape == (C, H, W, 4) and meta["isect_ids"].numel() == 0
    else:
        out = gsplat_amd.rasterization_2dgs(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"],
                                            leaves["colors"], sc["viewmats"], sc["Ks"], W, H, packed=packed,
                                            backgrounds=bg, render_mode="RGB+D", distloss=True)
        (out[0].sum() + out[1].sum() + out[2].sum() + out[4].sum()).backward()
        assert out[0].shape == (C, H, W, 4) and out[6]["isect_ids"].numel() == 0


def main() -> int:
    calls = install()
    bad = 0
    for fn in ("3dgs", "2dgs"):
        for packed in (False, True):
            n0 = len(calls)
            try:
                empty_scene(fn, packed)
                print(f"{fn} packed={packed}: OK ({len(calls) - n0} C-ABI calls: {sorted(set(calls[n0:]))})")
            except Exception as e:  # noqa: BLE001
                bad += 1
                print(f"{fn} packed={packed}: {type(e).__name__}: {str(e)[:300]}")
                for f in traceback.extract_tb(e.__traceback__)[-4:]:
                    print("    ", os.path.basename(f.filename), f.lineno, f.name)
    return bad


if __name__ == "__main__":
    sys.exit(main())
