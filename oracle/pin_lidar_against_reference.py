#!/usr/bin/env python
"""Golden vectors for the spinning-lidar pieces of 3DGUT from the reference's torch statements: the tile intersection
(gsplat/cuda/_torch_impl_lidar.py `_isect_tiles_lidar`) on small lidars built with the reference's own preprocessing
(gsplat/cuda/_lidar.py `compute_angles_to_columns_map`, `compute_tiling`). Writes tests/golden/lidar_ref.npz (the lidar's
tables, Gaussian boxes in angular pixels, the REFERENCE's counts / keys / ids); tests/test_gpu_lidar.py replays them on the GPU.
TEST INFRASTRUCTURE; run only where the reference checkout exists: python oracle/pin_lidar_against_reference.py"""
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

SCALE = 1024.0
# name: (rows, columns, elevation start / end, azimuth start / end of the base columns, clockwise, row offset amplitude, seed)
LIDARS = {
    "cw_120": (16, 120, 0.22, -0.21, 1.047, -1.047, True, 0.1, 3),
    "ccw_periodic": (12, 160, 0.25, -0.41, -3.0, 3.0, False, 0.2, 4),
    "ccw_90": (8, 64, 0.3, -0.3, -0.7, 0.8, False, 0.05, 5),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "lidar_ref.npz"))
    args = ap.parse_args()
    sys.path.insert(0, args.ref)
    sys.dont_write_bytecode = True
    import gsplat_amd.csrc_shim as shim

    sys.modules["gsplat.csrc"] = shim
    from gsplat.cuda import _lidar as L
    from gsplat.cuda._torch_impl_lidar import _isect_tiles_lidar

    gold = {}
    for name, (n_rows, n_cols, el0, el1, az0, az1, cw, amp, seed0) in LIDARS.items():
        direction = L.SpinningDirection.CLOCKWISE if cw else L.SpinningDirection.COUNTER_CLOCKWISE
        for seed in range(seed0, seed0 + 200):  # the reference's preprocessing asserts its own float bounds: take the first seed it accepts
            g = torch.Generator().manual_seed(seed)
            rows = torch.linspace(el0, el1, n_rows) + (torch.rand(n_rows, generator=g) - 0.5) * (abs(el1 - el0) / (n_rows - 1)) * 0.01
            cols = torch.linspace(az0, az1, n_cols) + (torch.rand(n_cols, generator=g) - 0.5) * (abs(az1 - az0) / (n_cols - 1)) * 0.01
            offs = (torch.rand(n_rows, generator=g) - 0.5) * 2 * amp
            try:
                params = L.RowOffsetStructuredSpinningLidarModelParameters(
                    row_elevations_rad=rows.float(), column_azimuths_rad=cols.float(), row_azimuth_offsets_rad=offs.float(),
                    spinning_frequency_hz=10.0, spinning_direction=direction)
                a2c = L.compute_angles_to_columns_map(params)
                tiling = L.compute_tiling(params, n_bins_elevation=4, max_pts_per_tile=64, resolution_elevation=64,
                                          densification_factor_azimuth=4)
                break
            except AssertionError as e:
                last = e
        else:
            raise last
        lidar = L.RowOffsetStructuredSpinningLidarModelParametersExt(params, a2c, tiling)
        # Gaussian boxes: means anywhere on the circle / around the vertical field of view, extents from a fraction of a tile to
        # more than the field of view (wrap-around, full cover, zero extent, outside the field of view)
        N, I = 600, 2
        az = (torch.rand(I, N, generator=g) * 2 - 1) * math.pi * SCALE
        el = (lidar.fov_vert_rad.start - torch.rand(I, N, generator=g) * lidar.fov_vert_rad.span * 1.4 + 0.2 * lidar.fov_vert_rad.span) * SCALE
        means2d = torch.stack([az, el], -1).float()
        radii = torch.stack([(torch.rand(I, N, generator=g) ** 3 * 2.2 * math.pi * SCALE).int(),
                             (torch.rand(I, N, generator=g) ** 2 * lidar.fov_vert_rad.span * SCALE).int()], -1)
        radii[:, ::17] = 0
        depths = torch.rand(I, N, generator=g) * 10 + 0.1
        depths[:, 1::50] = depths[:, ::50]  # equal depths: ties keep the row order
        tpg, ids, fl = _isect_tiles_lidar(lidar, means2d, radii, depths, sort=True)
        tpg_u, ids_u, fl_u = _isect_tiles_lidar(lidar, means2d, radii, depths, sort=False)
        print(f"{name:14s} fov az {lidar.fov_horiz_rad.span:.3f} rad ({'periodic' if lidar.fov_horiz_rad.span >= 2 * math.pi else 'open'}), "
              f"tiles {tiling.n_bins_azimuth} x {tiling.n_bins_elevation}, dense {tiling.cdf_resolution_azimuth} x "
              f"{tiling.cdf_resolution_elevation}; {int(tpg.sum())} intersections, rows with tiles {int((tpg > 0).sum())}/{tpg.numel()}, "
              f"max per row {int(tpg.max())}")
        assert int(tpg.sum()) > 1000
        rec = dict(row_elevations_rad=params.row_elevations_rad, column_azimuths_rad=params.column_azimuths_rad,
                   row_azimuth_offsets_rad=params.row_azimuth_offsets_rad, angles_to_columns_map=a2c,
                   cdf_elevation=tiling.cdf_elevation, cdf_dense_ray_mask=tiling.cdf_dense_ray_mask,
                   tiles_pack_info=tiling.tiles_pack_info, tiles_to_elements_map=tiling.tiles_to_elements_map,
                   means2d=means2d, radii=radii, depths=depths)
        for k, v in rec.items():
            gold[f"{name}.{k}"] = v.numpy()
        gold[f"{name}.scalars"] = np.array([lidar.fov_vert_rad.start, lidar.fov_vert_rad.span, lidar.fov_horiz_rad.start,
                                            lidar.fov_horiz_rad.span, lidar.fov_eps_rad, 0.0 if cw else 1.0, 10.0,
                                            tiling.n_bins_azimuth, tiling.n_bins_elevation], dtype=np.float64)
        gold[f"{name}.ref.tiles_per_gauss"], gold[f"{name}.ref.isect_ids"] = tpg.numpy(), ids.numpy()
        gold[f"{name}.ref.flatten_ids"] = fl.numpy()
        gold[f"{name}.ref.isect_ids_unsorted"], gold[f"{name}.ref.flatten_ids_unsorted"] = ids_u.numpy(), fl_u.numpy()
        # ---- the lidar as a CAMERA: unscented projection and element rays (global and rolling shutter) ------------------
        from gsplat.cuda._torch_impl_eval3d import _generate_rays
        from gsplat.cuda._torch_impl_ut import _fully_fused_projection_with_ut as ref_ut
        from gsplat.cuda._torch_lidars import _RowOffsetStructuredSpinningLidarModel
        from gsplat.cuda._wrapper import RollingShutterType

        Np, C = 500, 2
        r = torch.rand(Np, generator=g) * 7 + 2
        th_ = (torch.rand(Np, generator=g) * 2 - 1) * math.pi
        ph_ = (torch.rand(Np, generator=g) * 2 - 1) * 0.6
        pts = torch.stack([r * torch.cos(ph_) * torch.cos(th_), r * torch.cos(ph_) * torch.sin(th_), r * torch.sin(ph_)], -1)
        quats = torch.nn.functional.normalize(torch.randn(Np, 4, generator=g), dim=-1)
        scales = torch.rand(Np, 3, generator=g) * 0.15 + 0.02
        opac = torch.rand(Np, generator=g) * 0.9 + 0.1
        vm = torch.eye(4).repeat(C, 1, 1)
        for c in range(C):
            ax = torch.nn.functional.normalize(torch.randn(3, generator=g), dim=0)
            ang = 0.05 * (c + 1)
            K = torch.tensor([[0.0, -ax[2], ax[1]], [ax[2], 0.0, -ax[0]], [-ax[1], ax[0], 0.0]])
            vm[c, :3, :3] = torch.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)
            vm[c, :3, 3] = torch.randn(3, generator=g) * 0.2
        from pin_ut_rs_against_reference import end_poses

        vm1 = end_poses(vm, 7)
        Ks = torch.eye(3).repeat(C, 1, 1)
        import gsplat.cuda._torch_impl_ut as ut_mod

        class SpinningShutterLidar(_RowOffsetStructuredSpinningLidarModel):
            """The torch lidar with the shutter its CUDA twin hard-codes (Lidars.cuh:86: every lidar is a ROLLING sensor whose
            frame time is `shutter_relative_frame_time` of the angle map); the torch class leaves the base's GLOBAL default, which
            would project every sigma point with the start pose. The time function's own precondition assert (all angles inside
            the field of view) is lifted: the CUDA twin clamps the map index instead (Lidars.cuh:131-137), and so does the torch
            code below the assert."""

            def __init__(self, params):
                super().__init__(params)
                self.shutter_type = RollingShutterType.ROLLING_LEFT_TO_RIGHT

            def valid_sensor_angles(self, angles, *, scale=1):
                return torch.ones(angles.shape[:-1], dtype=torch.bool)

        for tag, rs, gz in (("global", 4, True), ("rs_distance", 0, False)):
            kw = dict(rolling_shutter=RollingShutterType(rs), viewmats_rs=vm1 if rs != 4 else None, global_z_order=gz)
            stock = ut_mod._RowOffsetStructuredSpinningLidarModel
            if rs != 4:
                ut_mod._RowOffsetStructuredSpinningLidarModel = SpinningShutterLidar
            try:
                ref = ref_ut(pts, quats, scales, opac, vm, Ks, n_cols, n_rows, camera_model="lidar", lidar_coeffs=lidar,
                             ut_params=torch.classes.gsplat.UnscentedTransformParameters(), **kw)
            finally:
                ut_mod._RowOffsetStructuredSpinningLidarModel = stock
            vis = (ref[0] > 0).all(-1)
            print(f"   projection {tag}: visible {int(vis.sum())}/{vis.numel()}")
            assert int(vis.sum()) > 30, (name, tag)
            for k, v in zip(("radii", "means2d", "depths", "conics"), ref[:4]):
                gold[f"{name}.proj_{tag}.{k}"] = v.numpy()
            cam = _RowOffsetStructuredSpinningLidarModel(lidar)
            rays = _generate_rays(cam, n_cols, n_rows, vm, vm1 if rs != 4 else None).reshape(C, n_rows, n_cols, 6)
            gold[f"{name}.rays_{tag}"] = rays.numpy()
        for k, v in dict(pts=pts, quats=quats, scales=scales, opac=opac, viewmats=vm, viewmats_rs=vm1, Ks=Ks).items():
            gold[f"{name}.cam.{k}"] = v.numpy()
    np.savez_compressed(args.out, **gold)
    print("wrote", args.out, os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()
