"""GPU: alternative kernel variants selected by environment switches must reproduce the default kernels.

GSX_RASTER3D_BWD=r — compositing backward with wave reductions (csrc/raster3d_bwd.hip) instead of the default transposed
per-Gaussian accumulation (variant T; used for <= 4 channels per launch, 16 x 16 tiles, no absgrad). The switch is read
once per process, so the other kernel runs in a subprocess on the same seeded scenes and its gradients are compared with
the default kernel's (scale-relative: the sums are taken in a different order). Both kernels are also what the
oracle-parity tests exercise: T through every default test, R through absgrad / wide-channel / small-tile cases."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

from _util import assert_grad_close

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    # N, C, W, H, sh_degree, render_mode, packed, backgrounds
    (6000, 2, 208, 144, 3, "RGB", False, True),
    (6000, 2, 208, 144, None, "RGB+ED", True, False),
    (3000, 1, 100, 70, None, "ED", False, False),      # 1 channel, image not a multiple of the tile
    (40000, 1, 320, 240, 0, "RGB", False, False),      # long per-tile lists: several batches and turns per tile
]

_SCRIPT = r'''
import sys, numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import gsplat_amd
from _util import make_scene
from test_gpu_variants import CASES, run_case
out = {}
for i, case in enumerate(CASES):
    for k, v in run_case(gsplat_amd, case).items():
        out[f"{i}_{k}"] = v
np.savez(sys.argv[1], **out)
'''


def run_case(G, case):
    from _util import make_scene

    N, C, W, H, sh_degree, mode, packed, with_bg = case
    sc, W, H = make_scene(N=N, C=C, width=W, height=H, seed=N % 97, sh_degree=sh_degree)
    nch = {"RGB": 3, "RGB+ED": 4, "ED": 1}[mode]
    g = torch.Generator().manual_seed(7)
    v_rc, v_ra = torch.randn(C, H, W, nch, generator=g).cuda(), torch.randn(C, H, W, 1, generator=g).cuda()
    bg = torch.rand(C, 3, generator=g).cuda() if with_bg else None
    names = ("means", "quats", "scales", "opacities", "colors")
    leaves = {k: sc[k].cuda().clone().requires_grad_(True) for k in names}
    rc, ra, _ = G.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"],
                                sc["viewmats"].cuda(), sc["Ks"].cuda(), W, H, sh_degree=sh_degree, packed=packed,
                                render_mode=mode, backgrounds=bg)
    ((rc * v_rc).sum() + (ra * v_ra).sum()).backward()
    out = {k: leaves[k].grad.cpu().numpy() for k in names if leaves[k].grad is not None}
    out["render"] = rc.detach().cpu().numpy()
    return out


@pytest.mark.parametrize("variant", ["r", "t", "w", "t+force", "w+force", "w+launch"])
def test_raster3d_bwd_variants_match_the_default(variant):
    """r: wave reductions; t: transposed per-Gaussian accumulation (four waves per tile); w: one wave per tile, four pixels per
    lane (csrc/raster3d_bwd.hip). Whichever is the default, the others must give the same gradients. "+force": the
    longest-first tile order also on these small images (it starts at 2048 tiles otherwise); "+launch": never."""
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    assert os.environ.get("GSX_RASTER3D_BWD", "") == "", "run this test with the default kernel selection"
    import gsplat_amd

    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "t.npz")
        code = _SCRIPT % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}
        kernel, _, order = variant.partition("+")
        env = dict(os.environ, GSX_RASTER3D_BWD=kernel)
        if order:
            env["GSX_RASTER3D_BWD_ORDER"] = order
        if os.environ.get("GSX_VARIANT_LIB"):  # an alternative build of the library for the variant side (A/B builds)
            env["GSPLAT_AMD_LIB"] = os.environ["GSX_VARIANT_LIB"]
        r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        alt = dict(np.load(path))
    for i, case in enumerate(CASES):
        ref = run_case(gsplat_amd, case)
        assert np.array_equal(alt[f"{i}_render"], ref["render"]), f"case {i}: the forward pass must not change"
        for k, v in ref.items():
            if k == "render":
                continue
            assert_grad_close(torch.from_numpy(alt[f"{i}_{k}"]), torch.from_numpy(v), rel=3e-4, max_bad_ratio=1e-5,
                              name=f"case {i} v_{k}")


@pytest.mark.parametrize("variant", ["w", "h"])
def test_raster3d_fwd_one_wave_per_tile_matches_the_default(variant):
    """GSX_RASTER3D_FWD=w / h: the forward with one wave per tile / per half tile (csrc/raster3d_fwd_w.hip; not the default) against
    the four-waves-per-tile kernel (csrc/raster3d_fwd.hip). Same staged form and the same per-pixel arithmetic in the same
    order: the renders must agree to rounding (a different cull never changes a result), and so must the gradients, which
    start from the forward's state."""
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    assert os.environ.get("GSX_RASTER3D_FWD", "") == "", "run this test with the default kernel selection"
    import gsplat_amd
    from _util import assert_close_ratio

    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "t.npz")
        code = _SCRIPT % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}
        r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True,
                           env=dict(os.environ, GSX_RASTER3D_FWD=variant), timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        alt = dict(np.load(path))
    for i, case in enumerate(CASES):
        ref = run_case(gsplat_amd, case)
        assert_close_ratio(torch.from_numpy(alt[f"{i}_render"]), torch.from_numpy(ref["render"]), 1e-6, 1e-6, max_bad_ratio=1e-5,
                           name=f"case {i} render")
        for k, v in ref.items():
            if k != "render":
                assert_grad_close(torch.from_numpy(alt[f"{i}_{k}"]), torch.from_numpy(v), rel=3e-4, max_bad_ratio=1e-5,
                                  name=f"case {i} v_{k}")


_SCRIPT_2DGS = r'''
import sys, numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import gsplat_amd
from test_gpu_variants import run_case_2dgs
np.savez(sys.argv[1], **run_case_2dgs(gsplat_amd))
'''


def run_case_2dgs(G):
    from _util import make_scene

    out = {}
    for tag, (N, C, W, H, mode, distloss, sh) in {"a": (6000, 2, 208, 144, "RGB+ED", True, 3), "b": (30000, 1, 160, 112, "RGB", False, None),
                                                  "c": (3000, 1, 100, 70, "D", True, None)}.items():
        sc, W, H = make_scene(N=N, C=C, width=W, height=H, seed=N % 89, sh_degree=sh, scale_range=(0.02, 0.15))
        names = ("means", "quats", "scales", "opacities", "colors")
        leaves = {k: sc[k].cuda().clone().requires_grad_(True) for k in names}
        rc, ra, rn, sn, rd, rm, _ = G.rasterization_2dgs(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"],
                                                        leaves["colors"], sc["viewmats"].cuda(), sc["Ks"].cuda(), W, H,
                                                        sh_degree=sh, render_mode=mode, distloss=distloss)
        g = torch.Generator().manual_seed(3)
        w = [torch.randn(t.shape, generator=g).cuda() for t in (rc, ra, rn, rd, rm)]
        ((rc * w[0]).sum() + (ra * w[1]).sum() + (rn * w[2]).sum() + (rd * w[3]).sum() + (rm * w[4]).sum()).backward()
        for k in names:
            if leaves[k].grad is not None:
                out[f"{tag}_{k}"] = leaves[k].grad.cpu().numpy()
        out[f"{tag}_render"] = rc.detach().cpu().numpy()
    return out


@pytest.mark.parametrize("variant", ["r", "w", "m"])
def test_raster2d_bwd_variants_match_the_default(variant):
    """The default is one wave per HALF tile (csrc/raster2d.hip: raster2d_bwd_w_kernel<.., 2>). GSX_RASTER2D_BWD=r (the four-wave
    reduction kernel), =w (the same kernel as the default, one wave per tile) and =m (csrc/raster2d_bwd_m.hip: the
    per-(tile, surfel) sums as one fp32-MFMA product per four surfels) - both measured and not the default - must give the
    gradients of the reduction kernel: RGB+ED with distortion loss and SH, long lists, a single depth channel."""
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    assert os.environ.get("GSX_RASTER2D_BWD", "") == ""
    import gsplat_amd

    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "t.npz")
        code = _SCRIPT_2DGS % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}
        r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True,
                           env=dict(os.environ, GSX_RASTER2D_BWD=variant, GSX_RASTER3D_BWD_ORDER="force"), timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        alt = dict(np.load(path))
    ref = run_case_2dgs(gsplat_amd)
    assert set(alt) == set(ref)
    for k, v in ref.items():
        if k.endswith("_render"):
            assert np.array_equal(alt[k], v), k
        else:
            # four pixels are summed per lane before ONE wave reduction (the default reduces per quadrant): another association
            # order of the same fp32 sums; the distortion terms of the depth-only case cancel strongly (1.3e-3 of scale on 3 of 12000)
            assert_grad_close(torch.from_numpy(alt[k]), torch.from_numpy(v), rel=2e-3, max_bad_ratio=1e-3, name=k)
