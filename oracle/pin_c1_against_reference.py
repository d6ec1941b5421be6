#!/usr/bin/env python
"""BASELINE.json configs[0] ("c1": 10k random Gaussians, 256 x 256 pinhole, SH degree 0 - the reference's own CPU-runnable
case) end to end: the reference's CPU path (gsplat/cuda/_torch_impl.py `_fully_fused_projection`,
`_spherical_harmonics`, `accumulate` with the restated nerfacc of pin_against_reference.py, sample lists from the oracle's
rasterize_to_indices - the one stage the reference has only in CUDA) against oracle/pipeline.py, and a compact fixture
tests/golden/c1_ref.npz (4096 sampled pixels + per-tile means of the REFERENCE image, not the whole 256 x 256 x 4 image).
    python oracle/pin_c1_against_reference.py [--ref /root/reference]
"""
import argparse
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from pin_against_reference import install_nerfacc_stub  # noqa: E402


def c1_scene(seed: int = 0):
    """SURVEY.md section 8(d) c1."""
    torch.manual_seed(seed)
    N, W, H = 10_000, 256, 256
    means = torch.rand(N, 3) * 2.0 - 1.0
    means[:, 2] = means[:, 2] * 2.0 + 4.0  # z in [2, 6]
    quats = torch.nn.functional.normalize(torch.randn(N, 4), dim=-1)
    scales = torch.rand(N, 3) * 0.04 + 0.01
    opacities = torch.rand(N) * 0.8 + 0.1
    colors = torch.rand(N, 1, 3)  # SH degree 0 coefficients
    viewmats = torch.eye(4)[None]
    Ks = torch.tensor([[[256.0, 0.0, 128.0], [0.0, 256.0, 128.0], [0.0, 0.0, 1.0]]])
    return dict(means=means, quats=quats, scales=scales, opacities=opacities, colors=colors, viewmats=viewmats, Ks=Ks), W, H


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "c1_ref.npz"))
    args = ap.parse_args()
    sys.path.insert(0, args.ref)
    sys.dont_write_bytecode = True
    install_nerfacc_stub()
    import gsplat  # noqa: F401
    from gsplat.cuda import _torch_impl as R
    from gsplat.cuda._math import _quat_scale_to_covar_preci
    from oracle import oracle as O
    from oracle.pipeline import rasterization_cpu

    sc, W, H = c1_scene()
    ts, tw, th = 16, 16, 16
    # ---- the reference's CPU path ----
    covars, _ = _quat_scale_to_covar_preci(sc["quats"], sc["scales"], True, False, triu=False)
    radii, means2d, depths, conics, _ = R._fully_fused_projection(sc["means"], covars, sc["viewmats"], sc["Ks"], W, H)
    dirs = sc["means"][None] - torch.linalg.inv(sc["viewmats"])[:, None, :3, 3]  # [C, N, 3] (unused by degree 0)
    sh = R._spherical_harmonics(0, dirs, sc["colors"])
    cols = torch.clamp_min(sh + 0.5, 0.0)
    op = sc["opacities"][None]
    # tile lists + sample lists: CUDA-only in the reference (isect_tiles / rasterize_to_indices) -> the pinned oracle stages
    rad_i = radii.to(torch.int32) if radii.dtype != torch.int32 else radii
    _, ids, fl = O.isect_tiles(means2d, rad_i, depths, ts, tw, th)
    off = O.isect_offset_encode(ids, 1, tw, th)
    g_ids, p_ids, i_ids = O.rasterize_to_indices(means2d, conics, op, W, H, ts, off, fl)
    ref_c, ref_a = R.accumulate(means2d, conics, op.contiguous(), cols.contiguous(), g_ids, p_ids, i_ids, W, H)
    # ---- the oracle pipeline ----
    out = rasterization_cpu(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["colors"], sc["viewmats"], sc["Ks"], W, H,
                            sh_degree=0, want_grads=False)
    d_c = float((out["render_colors"] - ref_c).abs().max())
    d_a = float((out["render_alphas"] - ref_a).abs().max())
    vis = int((radii > 0).all(-1).sum())
    print(f"c1: visible {vis}/10000, intersections {fl.numel()}, mean alpha {float(ref_a.mean()):.3f}; "
          f"oracle pipeline vs reference CPU path: max|d colour| {d_c:.2e}, max|d alpha| {d_a:.2e}")
    # the pipeline uses the exact ellipse tile test and the kernel's opacity-aware radii, the torch reference plain 3.33-sigma
    # boxes: both enumerate every contributing (pixel, Gaussian) pair, so the images agree
    assert d_c < 5e-5 and d_a < 5e-5
    g = torch.Generator().manual_seed(1)
    pix = torch.randperm(W * H, generator=g)[:4096]
    img = torch.cat([ref_c, ref_a], -1).reshape(W * H, 4)
    tiles = torch.cat([ref_c, ref_a], -1).reshape(th, ts, tw, ts, 4).mean(dim=(1, 3))
    np.savez_compressed(args.out, pixel_ids=pix.numpy(), pixels=img[pix].numpy(), tile_means=tiles.numpy(),
                        n_visible=np.array(vis), n_isects_aabb=np.array(fl.numel()))
    print("wrote", args.out, os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()
