#!/usr/bin/env python
"""Golden vectors for the ray generation of the from-world rasterizer (gsx_camera_rays): the world ray of every pixel centre for
each built camera model under a global or a rolling shutter, from the reference's torch statement of its camera models
(gsplat/cuda/_torch_cameras.py: `_BaseCameraModel.create(...)`, `image_point_to_world_ray_shutter_pose`) driven the way the
reference does it (gsplat/cuda/_torch_impl_eval3d.py:91-132 `_generate_rays`). Writes tests/golden/camera_rays_ref.npz (cameras
+ the REFERENCE's rays and validity); tests/test_gpu_eval3d.py replays them on the GPU.
TEST INFRASTRUCTURE; run only where the reference checkout exists: python oracle/pin_camera_rays_against_reference.py"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from pin_ut_against_reference import scene  # noqa: E402  (seeded cameras)
from pin_ut_rs_against_reference import FT, end_poses  # noqa: E402

C, W, H = 2, 48, 36
FT_BACKWARD = dict(FT, reference_poly=0)  # the pixel-distance -> angle polynomial is the calibrated one
# name: (seed, rolling shutter type 0..4 (4 = global), camera model, coefficients)
CASES = {
    "pinhole_global": (31, 4, "pinhole", {}),
    "pinhole_top_bottom": (32, 0, "pinhole", {}),
    "pinhole_left_right": (33, 1, "pinhole", {}),
    "pinhole_bottom_top": (34, 2, "pinhole", {}),
    "pinhole_right_left": (35, 3, "pinhole", {}),
    "opencv_global": (36, 4, "pinhole", dict(radial=[0.12, -0.06, 0.01, 0.02, -0.01, 0.004], tangential=[0.004, -0.003],
                                             thin_prism=[0.002, -0.001, 0.0015, 0.0005])),
    "opencv_radial4_top_bottom": (37, 0, "pinhole", dict(radial=[0.2, -0.05, 0.01, 0.0])),
    "opencv_strong": (38, 4, "pinhole", dict(radial=[-0.35, 0.12, 0.0, 0.0, 0.0, 0.0], tangential=[0.01, 0.008])),
    "ortho_global": (39, 4, "ortho", {}),
    "ortho_left_right": (40, 1, "ortho", {}),
    "fisheye_global": (41, 4, "fisheye", dict(radial=[-0.04, 0.012, -0.003, 0.0])),
    "fisheye_k4_bottom_top": (42, 2, "fisheye", dict(radial=[0.03, -0.01, 0.004, -0.0008])),
    "fisheye_plain": (43, 4, "fisheye", {}),
    "ftheta_forward_global": (44, 4, "ftheta", dict(ftheta=FT)),
    "ftheta_backward_right_left": (45, 3, "ftheta", dict(ftheta=FT_BACKWARD)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "camera_rays_ref.npz"))
    args = ap.parse_args()
    sys.path.insert(0, args.ref)
    sys.dont_write_bytecode = True
    import gsplat_amd.csrc_shim as shim

    sys.modules["gsplat.csrc"] = shim
    from gsplat.cuda._torch_cameras import _BaseCameraModel
    from gsplat.cuda._torch_impl_eval3d import _generate_rays
    from gsplat.cuda._wrapper import RollingShutterType

    gold = {}
    for name, (seed, rs, model, kw) in CASES.items():
        sc = scene(32, C, W, H, seed)
        vm0, Ks = sc["viewmats"], sc["Ks"].clone()
        if model == "ortho":
            Ks[:, 0, 0] *= 0.1
            Ks[:, 1, 1] *= 0.1
        vm1 = end_poses(vm0, seed) if rs != 4 else None
        coeffs = {k + "_coeffs": torch.tensor(kw[k]).repeat(C, 1) for k in ("radial", "tangential", "thin_prism") if k in kw}
        ft = kw.get("ftheta")
        cam_kw = dict(width=W, height=H, camera_model=model, principal_points=Ks[:, [0, 1], [2, 2]],
                      rs_type=RollingShutterType(rs), **coeffs)
        if ft is not None:
            cam_kw["ftheta_coeffs"] = torch.classes.gsplat.FThetaCameraDistortionParameters(
                ft["reference_poly"], ft["pixeldist_to_angle_poly"], ft["angle_to_pixeldist_poly"], ft["max_angle"],
                ft["linear_cde"])
        else:
            cam_kw["focal_lengths"] = Ks[:, [0, 1], [0, 1]]
        camera = _BaseCameraModel.create(**cam_kw)
        rays = _generate_rays(camera, W, H, vm0, vm1).reshape(C, H, W, 6)
        valid = rays[..., 3:].abs().sum(-1) > 0
        norm = rays[..., 3:].norm(dim=-1)[valid]
        assert float((norm - 1).abs().max()) < 1e-5, name
        print(f"{name:30s} valid pixels {int(valid.sum()):5d}/{valid.numel()}  direction spread "
              f"{float(rays[..., 3:][valid].std(0).mean()):.3f}")
        assert int(valid.sum()) > 0.5 * valid.numel(), name
        gold[f"{name}.viewmats"], gold[f"{name}.Ks"] = vm0.numpy(), Ks.numpy()
        if vm1 is not None:
            gold[f"{name}.viewmats_rs"] = vm1.numpy()
        for k, v in coeffs.items():
            gold[f"{name}.{k}"] = v.numpy()
        if ft is not None:
            gold[f"{name}.ftheta"] = np.array([ft["reference_poly"]] + ft["pixeldist_to_angle_poly"] + ft["angle_to_pixeldist_poly"]
                                              + [ft["max_angle"]] + ft["linear_cde"], dtype=np.float64)
        gold[f"{name}.meta"] = np.array([rs, {"pinhole": 0, "ortho": 1, "fisheye": 2, "ftheta": 3}[model], W, H])
        gold[f"{name}.ref.rays"] = rays.numpy()
    np.savez_compressed(args.out, **gold)
    print("wrote", args.out, os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()
