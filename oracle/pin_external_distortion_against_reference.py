#!/usr/bin/env python
"""Golden vectors for the external (windshield) distortion model (gsx_distort_camera_rays, gsx_eval_bivariate_poly and the
`ext` record of gsx_camera_rays_ext / gsx_project_ut_ext_fwd) from the reference's Python statement of it
(gsplat/cuda/_torch_external_distortion.py: `ref_eval_bivariate_poly`, `ref_distort_camera_ray` - the functions the reference's
own tests/test_external_distortion.py compares its CUDA kernels with). Writes tests/golden/external_distortion_ref.npz
(polynomials, inputs, the REFERENCE's outputs); tests/test_gpu_eval3d.py replays them on the GPU.
TEST INFRASTRUCTURE; run only where the reference checkout exists: python oracle/pin_external_distortion_against_reference.py"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "external_distortion_ref.npz"))
    args = ap.parse_args()
    sys.path.insert(0, args.ref)
    sys.dont_write_bytecode = True
    import gsplat_amd.csrc_shim as shim

    sys.modules["gsplat.csrc"] = shim
    from gsplat.cuda import _torch_external_distortion as R

    g = torch.Generator().manual_seed(77)
    gold = {}
    for order in range(6):
        n = R.num_coeffs_for_order(order)
        # a perturbed identity: phi' = phi + small terms, theta' = theta + small terms (what a windshield does)
        h = (torch.randn(n, generator=g) * 0.02).tolist()
        v = (torch.randn(n, generator=g) * 0.02).tolist()
        if order >= 1:
            h[1] += 1.0
            v[order + 1] += 1.0
        xs = (torch.rand(200, generator=g) * 1.6 - 0.8).tolist()
        ys = (torch.rand(200, generator=g) * 1.6 - 0.8).tolist()
        vals = [R.ref_eval_bivariate_poly(h, order, x, y) for x, y in zip(xs, ys)]
        rays = torch.randn(300, 3, generator=g)
        rays[:, 2] = rays[:, 2].abs() + 0.2
        rays[::7, 2] *= -1.0   # behind the camera: z keeps its sign
        rays[5] = 0.0          # the zero ray passes through
        out = [R.ref_distort_camera_ray(tuple(r.tolist()), h, v, order, order) for r in rays]
        gold[f"o{order}.h"], gold[f"o{order}.v"] = np.array(h, dtype=np.float64), np.array(v, dtype=np.float64)
        gold[f"o{order}.x"], gold[f"o{order}.y"] = np.array(xs, dtype=np.float32), np.array(ys, dtype=np.float32)
        gold[f"o{order}.ref.poly"] = np.array(vals, dtype=np.float64)
        gold[f"o{order}.rays"] = rays.numpy()
        gold[f"o{order}.ref.distorted"] = np.array(out, dtype=np.float64)
        print(f"order {order}: {n} coefficients, max |distorted - normalised input| "
              f"{float(np.abs(np.array(out) - (rays / rays.norm(dim=-1, keepdim=True).clamp_min(1e-12)).numpy())[np.arange(300) != 5].max()):.3f}")
    np.savez_compressed(args.out, **gold)
    print("wrote", args.out, os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()
