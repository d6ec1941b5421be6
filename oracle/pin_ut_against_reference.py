#!/usr/bin/env python
"""Pin oracle/ut.py against the reference's own torch statement of the UT projection
(gsplat/cuda/_torch_impl_ut.py:306-644 `_fully_fused_projection_with_ut`, camera models gsplat/cuda/_torch_cameras.py) and
write tests/golden/ut_ref.npz (inputs + the REFERENCE's outputs).

The reference needs its compiled extension only for the parameter records (`torch.classes.gsplat.*`); those come from this
backend's libgsplat_amd_torch.so, installed as `gsplat.csrc` (INTEGRATION.md route A) - everything else in that module is
plain torch and runs on the CPU. Run only where the reference checkout exists:
    python oracle/pin_ut_against_reference.py [--ref /root/reference]
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


def scene(N, C, W, H, seed):
    g = torch.Generator().manual_seed(seed)
    f = 0.8 * W
    z = torch.rand(N, generator=g) * 6.0 + 0.3
    means = torch.stack([(torch.rand(N, generator=g) - 0.5) * 1.6 * W / f * z, (torch.rand(N, generator=g) - 0.5) * 1.6 * H / f * z, z], -1)
    means[::23, 2] *= -1.0  # behind the camera
    quats = torch.randn(N, 4, generator=g)
    quats[5] = 0.0  # no orientation: culled
    scales = torch.exp(torch.randn(N, 3, generator=g) * 0.6 + math.log(0.06))
    scales[9, 1] = 0.0  # degenerate axis: culled
    opacities = torch.rand(N, generator=g)
    opacities[::31] = 0.002  # below 1/255
    viewmats = torch.eye(4).repeat(C, 1, 1)
    for c in range(C):
        a = 0.08 * c
        viewmats[c, :3, :3] = torch.tensor([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]])
        viewmats[c, :3, 3] = torch.tensor([0.1 * c, -0.05 * c, 0.02 * c])
    Ks = torch.tensor([[f, 0.0, W / 2 + 1.5], [0.0, 0.9 * f, H / 2 - 2.0], [0.0, 0.0, 1.0]]).repeat(C, 1, 1)
    return dict(means=means, quats=quats, scales=scales, opacities=opacities, viewmats=viewmats, Ks=Ks)


CASES = {
    # name: (scene seed, kwargs of the projection)
    "pinhole": (1, dict()),
    "pinhole_all_valid": (2, dict(require_all_sigma_points_valid=True, in_image_margin_factor=0.05)),
    "pinhole_comp_clip": (3, dict(calc_compensations=True, radius_clip=2.0, eps2d=0.1, near_plane=0.5, far_plane=5.0)),
    "pinhole_no_opacity": (4, dict(use_opacities=False, alpha=0.5, beta=1.0, kappa=0.5)),
    "opencv_full": (5, dict(radial=[0.12, -0.06, 0.01, 0.02, -0.01, 0.004], tangential=[0.004, -0.003],
                            thin_prism=[0.002, -0.001, 0.0015, 0.0005])),
    "opencv_radial4": (6, dict(radial=[-0.2, 0.05, 0.0, 0.0], require_all_sigma_points_valid=True)),
    "opencv_strong": (7, dict(radial=[-0.45, 0.1, 0.0, 0.0, 0.0, 0.0], tangential=[0.01, 0.01])),  # icD < 0.8 at the rim
    "ortho": (8, dict(camera_model="ortho")),
    "fisheye_plain": (9, dict(camera_model="fisheye")),
    "fisheye_k": (10, dict(camera_model="fisheye", radial=[-0.04, 0.012, -0.003, 0.0])),   # k4 = 0: cubic branch
    "fisheye_k4": (11, dict(camera_model="fisheye", radial=[0.05, -0.02, 0.004, -0.0015],  # k4 != 0: Newton branch
                            require_all_sigma_points_valid=True)),
    "fisheye_tight": (12, dict(camera_model="fisheye", radial=[-0.35, 0.0, 0.0, 0.0])),     # monotonic only up to ~0.98 rad
    # f-theta (oracle only so far): pixel distance ~ 76.8 theta - 2 theta^3; both polynomial directions as the reference one
    "ftheta_forward": (13, dict(camera_model="ftheta", ftheta=dict(
        reference_poly=1, pixeldist_to_angle_poly=[0.0, 1 / 76.8, 0.0, 2 / 76.8 ** 4, 0.0, 0.0],
        angle_to_pixeldist_poly=[0.0, 76.8, 0.0, -2.0, 0.0, 0.0], max_angle=1.2, linear_cde=[1.0, 0.0, 0.0]))),
    "ftheta_inverse": (14, dict(camera_model="ftheta", require_all_sigma_points_valid=True, ftheta=dict(
        reference_poly=0, pixeldist_to_angle_poly=[0.0, 1 / 76.8, 0.0, 2 / 76.8 ** 4, 0.0, 0.0],
        angle_to_pixeldist_poly=[0.0, 76.8, 0.0, -2.0, 0.0, 0.0], max_angle=0.9, linear_cde=[1.002, 0.001, -0.0015]))),
}
N, C, W, H = 400, 2, 96, 64


def split(kw):
    kw = dict(kw)
    ut = dict(alpha=kw.pop("alpha", 0.1), beta=kw.pop("beta", 2.0), kappa=kw.pop("kappa", 0.0),
              in_image_margin_factor=kw.pop("in_image_margin_factor", 0.1),
              require_all_sigma_points_valid=kw.pop("require_all_sigma_points_valid", False))
    dist = {k: kw.pop(k, None) for k in ("radial", "tangential", "thin_prism")}
    use_op = kw.pop("use_opacities", True)
    return kw, ut, dist, use_op  # kw may still hold camera_model / ftheta / eps2d / near_plane / ...


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "ut_ref.npz"))
    args = ap.parse_args()
    sys.path.insert(0, args.ref)
    sys.dont_write_bytecode = True
    import gsplat_amd.csrc_shim as shim

    sys.modules["gsplat.csrc"] = shim
    from gsplat.cuda._torch_impl_ut import _fully_fused_projection_with_ut as ref_ut
    from oracle import ut as O

    gold, worst = {}, {}
    for name, (seed, kw) in CASES.items():
        sc = scene(N, C, W, H, seed)
        kw, ut, dist, use_op = split(kw)
        cam = {k + "_coeffs": (None if v is None else torch.tensor(v).repeat(C, 1)) for k, v in dist.items()}
        if kw.get("camera_model") == "ortho":
            sc["Ks"][:, 0, 0] *= 0.1
            sc["Ks"][:, 1, 1] *= 0.1
        op = sc["opacities"] if use_op else None
        ft = kw.get("ftheta")
        kw_ref = {k: v for k, v in kw.items() if k != "ftheta"}
        if ft is not None:
            kw_ref["ftheta_coeffs"] = torch.classes.gsplat.FThetaCameraDistortionParameters(
                ft["reference_poly"], ft["pixeldist_to_angle_poly"], ft["angle_to_pixeldist_poly"], ft["max_angle"],
                ft["linear_cde"])
        ref = ref_ut(sc["means"], sc["quats"], sc["scales"], op, sc["viewmats"], sc["Ks"], W, H,
                     ut_params=torch.classes.gsplat.UnscentedTransformParameters(**ut), **cam, **kw_ref)
        got = O.fully_fused_projection_with_ut(sc["means"], sc["quats"], sc["scales"], op, sc["viewmats"], sc["Ks"], W, H,
                                               **ut, **cam, **kw)
        vis_r, vis_g = (ref[0] > 0).all(-1), (got[0] > 0).all(-1)
        flips = int((vis_r != vis_g).sum())
        both = vis_r & vis_g
        dr = int((ref[0] - got[0]).abs()[both].max()) if both.any() else 0
        dm = float((ref[1] - got[1]).abs()[both].max()) if both.any() else 0.0
        dc = float(((ref[3] - got[3]).abs() / (ref[3].abs() + 1e-3))[both].max()) if both.any() else 0.0
        worst[name] = (int(vis_r.sum()), flips, dr, dm, dc)
        print(f"{name:20s} visible {int(vis_r.sum()):4d}/{vis_r.numel()}  validity flips {flips}  max |d radii| {dr}  "
              f"max |d means2d| {dm:.2e}  max rel d conics {dc:.2e}")
        # the reference's own CUDA-vs-torch tolerances for this op (tests/test_basic.py:838-960, global shutter): validity
        # mismatches < 0.1 %, radii atol 2, means2d rtol 2e-3 / atol 5e-2 - with alpha = 0.1 the centre weight is ~ -99, so
        # any two fp32 evaluation orders differ by ~1e-3 px
        assert flips <= max(1, vis_r.numel() // 1000) and dr <= 1 and dm < 2e-2 and dc < 3e-2, name
        for k, v in sc.items():
            gold[f"{name}.{k}"] = v.numpy()
        for k, v in zip(("radii", "means2d", "depths", "conics", "compensations"), ref):
            if v is not None:
                gold[f"{name}.ref.{k}"] = v.numpy()
    # the per-camera angle limit of the fisheye model (every branch: linear, quadratic, Cardano, three roots, Newton)
    from gsplat.cuda._torch_cameras import _OpenCVFisheyeCameraModel
    from gsplat.cuda._wrapper import RollingShutterType

    g = torch.Generator().manual_seed(3)
    ks = torch.cat([
        torch.tensor([[0, 0, 0, 0.0], [-0.04, 0.012, -0.003, 0], [0.05, -0.02, 0.004, -0.0015], [-0.35, 0, 0, 0],
                      [0.1, 0, 0, 0], [0, -0.2, 0, 0], [0.02, 0.01, -0.05, 0], [0, 0, 0, -0.01], [-0.1, -0.1, 0, 0]]),
        torch.randn(40, 4, generator=g) * torch.tensor([0.2, 0.1, 0.05, 0.02]),
        torch.cat([torch.randn(20, 3, generator=g) * torch.tensor([0.2, 0.1, 0.05]), torch.zeros(20, 1)], 1)])
    Kf = torch.tensor([[76.8, 0, 1000.0], [0, 69.1, 800.0], [0, 0, 1]]).repeat(len(ks), 1, 1)
    cam = _OpenCVFisheyeCameraModel(focal_lengths=torch.stack([Kf[:, 0, 0], Kf[:, 1, 1]], -1), principal_points=Kf[:, :2, 2],
                                    width=2000, height=1600, rs_type=RollingShutterType.GLOBAL, radial_coeffs=ks)
    mine = O.fisheye_max_angle(ks, Kf[:, 0, 0], Kf[:, 1, 1], Kf[:, 0, 2], Kf[:, 1, 2], 2000, 1600)
    fin = torch.isfinite(cam.max_angle)
    assert bool((torch.isfinite(mine) == fin).all()) and float((mine - cam.max_angle)[fin].abs().max()) < 1e-5
    gold["fisheye_limit.k"], gold["fisheye_limit.Ks"] = ks.numpy(), Kf.numpy()
    gold["fisheye_limit.ref.max_angle"] = cam.max_angle.numpy()
    print("fisheye angle limit: %d coefficient sets, oracle == reference" % len(ks))
    np.savez_compressed(args.out, **gold)
    print("wrote", args.out, os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()
