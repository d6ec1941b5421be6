"""CPU: host-side helpers and the algebra the kernels rely on (no kernels run here)."""
import numpy as np
import pytest
import torch


def test_world_to_cam_matches_definition():
    import gsplat_amd

    g = torch.Generator().manual_seed(0)
    means, covars = torch.randn(2, 7, 3, generator=g), torch.randn(2, 7, 3, 3, generator=g)
    covars = covars @ covars.transpose(-1, -2)
    viewmats = torch.randn(2, 3, 4, 4, generator=g)
    mc, cc = gsplat_amd.world_to_cam(means, covars, viewmats)
    assert mc.shape == (2, 3, 7, 3) and cc.shape == (2, 3, 7, 3, 3)
    R, t = viewmats[1, 2, :3, :3], viewmats[1, 2, :3, 3]
    assert torch.allclose(mc[1, 2, 4], R @ means[1, 4] + t, atol=1e-5)
    assert torch.allclose(cc[1, 2, 4], R @ covars[1, 4] @ R.T, atol=1e-4)


def test_feature_probes_follow_build_config():
    import gsplat_amd

    cfg = gsplat_amd.build_config()
    assert set(cfg) == {"3dgs", "2dgs", "3dgut", "adam", "reloc", "losses", "camera_wrappers"}  # ext.cpp:83-97
    assert gsplat_amd.has_3dgs() and gsplat_amd.has_2dgs() and gsplat_amd.has_adam() and gsplat_amd.has_reloc()
    assert gsplat_amd.has_3dgut() and not gsplat_amd.has_losses() and not gsplat_amd.has_camera_wrappers()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_tile_origin_moments_give_the_mean_relative_moments(seed):
    """The compositing backward (variant T, csrc/raster3d_bwd.hip) sums w, w u, w v, w u^2, w u v, w v^2 over the pixels of a
    tile with (u, v) measured from the TILE origin, and turns them into the moments of d = mean - pixel = a - (u, v) once per
    (tile, Gaussian). This is that conversion, in float32 like the kernel, against the direct sums in float64."""
    rng = np.random.default_rng(seed)
    u, v = np.meshgrid(np.arange(16, dtype=np.float32), np.arange(16, dtype=np.float32))
    w = rng.normal(size=(16, 16)).astype(np.float32) * np.exp(-rng.uniform(0, 4, size=(16, 16))).astype(np.float32)
    for ax, ay in ((3.7, 9.2), (-250.5, 40.25), (700.0, -320.0)):
        ax32, ay32 = np.float32(ax), np.float32(ay)
        S0, Su, Sv = w.sum(dtype=np.float32), (w * u).sum(dtype=np.float32), (w * v).sum(dtype=np.float32)
        Suu, Suv, Svv = (w * u * u).sum(dtype=np.float32), (w * u * v).sum(dtype=np.float32), (w * v * v).sum(dtype=np.float32)
        got = dict(x=ax32 * S0 - Su, y=ay32 * S0 - Sv, xx=ax32 * (ax32 * S0 - np.float32(2) * Su) + Suu,
                   xy=ax32 * (ay32 * S0 - Sv) - ay32 * Su + Suv, yy=ay32 * (ay32 * S0 - np.float32(2) * Sv) + Svv)
        w64, dx, dy = w.astype(np.float64), ax - u.astype(np.float64), ay - v.astype(np.float64)
        want = dict(x=(w64 * dx).sum(), y=(w64 * dy).sum(), xx=(w64 * dx * dx).sum(), xy=(w64 * dx * dy).sum(),
                    yy=(w64 * dy * dy).sum())
        scale = dict(x=(np.abs(w64) * np.abs(dx)).sum(), y=(np.abs(w64) * np.abs(dy)).sum(),
                     xx=(np.abs(w64) * dx * dx).sum(), xy=(np.abs(w64) * np.abs(dx * dy)).sum(),
                     yy=(np.abs(w64) * dy * dy).sum())
        for k in want:  # error relative to the size of the terms that are summed (what an fp32 direct sum would also see)
            assert abs(float(got[k]) - want[k]) <= 2e-5 * scale[k] + 1e-6, (k, ax, ay, float(got[k]), want[k])
