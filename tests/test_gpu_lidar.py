"""GPU: the spinning-lidar pieces of 3DGUT against golden vectors of the reference's torch statements
(tests/golden/lidar_ref.npz, oracle/pin_lidar_against_reference.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"


@pytest.fixture(scope="module")
def G():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    import gsplat_amd
    from gsplat_amd import _ops  # noqa: F401

    return gsplat_amd


def lidar_from_golden(gold, name, dev=DEV):
    """The custom-class record (what the reference's `to_cpp()` builds) from the tables stored in the golden file."""
    c = torch.classes.gsplat
    t = lambda k: torch.from_numpy(gold[f"{name}.{k}"]).to(dev)  # noqa: E731
    fv0, fvs, fh0, fhs, eps, ccw, hz, nb_az, nb_el = (float(v) for v in gold[f"{name}.scalars"])
    return c.RowOffsetStructuredSpinningLidarModelParametersExt(
        t("row_elevations_rad"), t("column_azimuths_rad"), t("row_azimuth_offsets_rad"), int(ccw), hz, c.FOV(fv0, fvs),
        c.FOV(fh0, fhs), eps, t("angles_to_columns_map"), int(nb_az), int(nb_el), t("cdf_elevation"), t("cdf_dense_ray_mask"),
        t("tiles_pack_info"), t("tiles_to_elements_map"))


@pytest.mark.parametrize("name", ["cw_120", "ccw_periodic", "ccw_90"])
def test_lidar_tile_intersection_equals_reference(G, name):
    """gsplat::intersect_tile_lidar (gsx_isect_lidar_{count,emit} + the per-tile sort) against the reference's torch statement,
    EXACTLY: counts per Gaussian, the unsorted (image | tile | depth) keys in emission order, and the sorted lists - boxes that
    wrap around the azimuth seam, cover the whole field of view, lie outside it, have no extent, tie in depth; an open and a
    periodic field of view, both spinning directions."""
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "lidar_ref.npz")))
    lidar = lidar_from_golden(gold, name)
    t = lambda k: torch.from_numpy(gold[f"{name}.{k}"]).to(DEV)  # noqa: E731
    means2d, radii, depths = t("means2d"), t("radii"), t("depths")
    tpg, ids, fl = torch.ops.gsplat.intersect_tile_lidar(lidar, means2d, radii, depths, None, None, None, False, False)
    assert torch.equal(tpg.cpu(), torch.from_numpy(gold[f"{name}.ref.tiles_per_gauss"]))
    assert torch.equal(ids.cpu(), torch.from_numpy(gold[f"{name}.ref.isect_ids_unsorted"]))
    assert torch.equal(fl.cpu(), torch.from_numpy(gold[f"{name}.ref.flatten_ids_unsorted"]))
    tpg, ids, fl = torch.ops.gsplat.intersect_tile_lidar(lidar, means2d, radii, depths, None, None, None, True, False)
    assert torch.equal(ids.cpu(), torch.from_numpy(gold[f"{name}.ref.isect_ids"]))
    ref_fl = torch.from_numpy(gold[f"{name}.ref.flatten_ids"])
    same = fl.cpu() == ref_fl  # equal keys (same tile, same depth bits) may come in either order out of torch.sort
    assert bool(same.all()) or bool((ids.cpu()[~same][:, None] == ids.cpu()[~same][None, :]).any(1).all())
    # float radii and packed rows give the same lists
    I, N = means2d.shape[0], means2d.shape[1]
    img = torch.arange(I, device=DEV).repeat_interleave(N)
    gid = torch.arange(N, device=DEV).repeat(I)
    tpg_p, ids_p, fl_p = torch.ops.gsplat.intersect_tile_lidar(lidar, means2d.reshape(-1, 2), radii.reshape(-1, 2).float(),
                                                               depths.reshape(-1), img, gid, I, True, False)
    assert torch.equal(tpg_p, tpg.reshape(-1)) and torch.equal(ids_p, ids) and torch.equal(fl_p, fl)
    offsets = torch.ops.gsplat.intersect_offset(ids, I, int(lidar.n_bins_azimuth), int(lidar.n_bins_elevation))
    assert offsets.shape == (I, int(lidar.n_bins_elevation), int(lidar.n_bins_azimuth))
