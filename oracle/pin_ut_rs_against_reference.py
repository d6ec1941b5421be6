#!/usr/bin/env python
"""Golden vectors for the ROLLING-SHUTTER / Euclidean-depth UT projection, from the reference's own torch statement
(gsplat/cuda/_torch_impl_ut.py:306-644 `_fully_fused_projection_with_ut` with `rolling_shutter`, `viewmats_rs`,
`global_z_order`; camera models and the rolling-shutter fixed point in gsplat/cuda/_torch_cameras.py:424-660, 2163-2210).
Writes tests/golden/ut_rs_ref.npz = inputs + the REFERENCE's outputs; tests/test_gpu_ut.py replays them on the kernel
(gsx_project_ut_rs_fwd). TEST INFRASTRUCTURE; run only where the reference checkout exists:
    python oracle/pin_ut_rs_against_reference.py [--ref /root/reference]
The parameter records (`torch.classes.gsplat.*`) come from this backend's libgsplat_amd_torch.so installed as `gsplat.csrc`;
everything else in the reference module is plain torch on the CPU."""
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

from pin_ut_against_reference import scene  # noqa: E402  (the same seeded scenes as the global-shutter table)

N, C, W, H = 400, 2, 96, 64
FT = dict(reference_poly=1, pixeldist_to_angle_poly=[0.0, 1 / 76.8, 0.0, 2 / 76.8 ** 4, 0.0, 0.0],
          angle_to_pixeldist_poly=[0.0, 76.8, 0.0, -2.0, 0.0, 0.0], max_angle=1.2, linear_cde=[1.0, 0.0, 0.0])
# name: (scene seed, rolling shutter type 0..4 (4 = global), global_z_order, kwargs)
CASES = {
    "rs_top_bottom_pinhole": (21, 0, True, dict()),
    "rs_left_right_pinhole_all_valid": (22, 1, True, dict(require_all_sigma_points_valid=True)),
    "rs_bottom_top_opencv": (23, 2, True, dict(radial=[0.12, -0.06, 0.01, 0.02, -0.01, 0.004], tangential=[0.004, -0.003],
                                                thin_prism=[0.002, -0.001, 0.0015, 0.0005])),
    "rs_right_left_fisheye": (24, 3, True, dict(camera_model="fisheye", radial=[-0.04, 0.012, -0.003, 0.0])),
    "rs_top_bottom_ortho": (25, 0, True, dict(camera_model="ortho")),
    "rs_left_right_ftheta": (26, 1, True, dict(camera_model="ftheta", ftheta=FT)),
    "rs_top_bottom_distance": (27, 0, False, dict(calc_compensations=True)),
    "global_distance_pinhole": (28, 4, False, dict()),
    "global_distance_ftheta": (29, 4, False, dict(camera_model="ftheta", ftheta=FT, near_plane=0.5, far_plane=5.0)),
}


def end_poses(viewmats, seed):
    """The pose at the end of the frame: the start pose moved by a small rotation + translation (a few pixels of motion)."""
    g = torch.Generator().manual_seed(1000 + seed)
    out = viewmats.clone()
    for c in range(viewmats.shape[0]):
        ax = torch.randn(3, generator=g)
        ax = ax / ax.norm()
        ang = 0.010 + 0.005 * c
        K = torch.tensor([[0.0, -ax[2], ax[1]], [ax[2], 0.0, -ax[0]], [-ax[1], ax[0], 0.0]])
        dR = torch.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)
        out[c, :3, :3] = dR @ viewmats[c, :3, :3]
        out[c, :3, 3] = dR @ viewmats[c, :3, 3] + torch.tensor([0.012, -0.008, 0.006]) * (1 + c)
    return out


def split(kw):
    kw = dict(kw)
    ut = dict(alpha=0.1, beta=2.0, kappa=0.0, in_image_margin_factor=0.1,
              require_all_sigma_points_valid=kw.pop("require_all_sigma_points_valid", False))
    dist = {k: kw.pop(k, None) for k in ("radial", "tangential", "thin_prism")}
    return kw, ut, dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "ut_rs_ref.npz"))
    args = ap.parse_args()
    sys.path.insert(0, args.ref)
    sys.dont_write_bytecode = True
    import gsplat_amd.csrc_shim as shim

    sys.modules["gsplat.csrc"] = shim
    from gsplat.cuda._torch_impl_ut import _fully_fused_projection_with_ut as ref_ut
    from gsplat.cuda._wrapper import RollingShutterType

    gold = {}
    for name, (seed, rs, gz, kw) in CASES.items():
        sc = scene(N, C, W, H, seed)
        kw, ut, dist = split(kw)
        cam = {k + "_coeffs": (None if v is None else torch.tensor(v).repeat(C, 1)) for k, v in dist.items()}
        if kw.get("camera_model") == "ortho":
            sc["Ks"][:, 0, 0] *= 0.1
            sc["Ks"][:, 1, 1] *= 0.1
        ft = kw.pop("ftheta", None)
        if ft is not None:
            kw["ftheta_coeffs"] = torch.classes.gsplat.FThetaCameraDistortionParameters(
                ft["reference_poly"], ft["pixeldist_to_angle_poly"], ft["angle_to_pixeldist_poly"], ft["max_angle"],
                ft["linear_cde"])
        vm1 = end_poses(sc["viewmats"], seed) if rs != 4 else None
        ref = ref_ut(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["viewmats"], sc["Ks"], W, H,
                     ut_params=torch.classes.gsplat.UnscentedTransformParameters(**ut), rolling_shutter=RollingShutterType(rs),
                     viewmats_rs=vm1, global_z_order=gz, **cam, **kw)
        vis = (ref[0] > 0).all(-1)
        if rs != 4:  # the motion must matter: the same scene through a global shutter differs by pixels
            glob = ref_ut(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["viewmats"], sc["Ks"], W, H,
                          ut_params=torch.classes.gsplat.UnscentedTransformParameters(**ut), global_z_order=gz, **cam, **kw)
            both = vis & (glob[0] > 0).all(-1)
            shift = float((ref[1] - glob[1])[both].abs().max()) if both.any() else 0.0
            assert shift > 0.5, (name, shift)
        else:
            shift = 0.0
        print(f"{name:34s} visible {int(vis.sum()):4d}/{vis.numel()}  max pixel shift against a global shutter {shift:.2f}")
        assert int(vis.sum()) > 40, name
        for k, v in sc.items():
            gold[f"{name}.{k}"] = v.numpy()
        if vm1 is not None:
            gold[f"{name}.viewmats_rs"] = vm1.numpy()
        for k, v in zip(("radii", "means2d", "depths", "conics", "compensations"), ref):
            if v is not None:
                gold[f"{name}.ref.{k}"] = v.numpy()
    np.savez_compressed(args.out, **gold)
    print("wrote", args.out, os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()
