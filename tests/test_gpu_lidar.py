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


@pytest.mark.parametrize("name", ["cw_120", "ccw_periodic", "ccw_90"])
@pytest.mark.parametrize("tag,rs,gz", [("global", 4, True), ("rs_distance", 0, False)])
def test_lidar_unscented_projection_and_rays_match_reference(G, name, tag, rs, gz):
    """The lidar as a camera of the two 3DGUT kernels against the reference's torch statements: gsx_project_ut_lidar_fwd (image
    points = (azimuth, elevation) * 1024, field-of-view validity, radial culling / depth with global_z_order=False, the rolling
    shutter's time read off the angle map) and gsx_lidar_rays (one world ray per element)."""
    from gsplat_amd import _ops

    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "lidar_ref.npz")))
    lidar = lidar_from_golden(gold, name)
    t = lambda k: torch.from_numpy(gold[f"{name}.cam.{k}"]).to(DEV)  # noqa: E731
    vm1 = t("viewmats_rs") if rs != 4 else None
    radii, m2, dep, con, _ = G.fully_fused_projection_with_ut(
        t("pts"), t("quats"), t("scales"), t("opac"), t("viewmats"), t("Ks"), int(lidar.column_azimuths_rad.shape[0]),
        int(lidar.row_elevations_rad.shape[0]), camera_model="lidar", lidar_coeffs=lidar, rolling_shutter=rs, viewmats_rs=vm1,
        global_z_order=gz)
    ref = {k: torch.from_numpy(gold[f"{name}.proj_{tag}.{k}"]) for k in ("radii", "means2d", "depths", "conics")}
    vis, vis_r = (radii.cpu() > 0).all(-1), (ref["radii"] > 0).all(-1)
    assert float((vis == vis_r).float().mean()) > 0.995  # a sigma point within rounding of the field of view's edge may flip a row
    both = vis & vis_r
    assert int(both.sum()) > 30
    assert int((radii.cpu()[both] - ref["radii"][both]).abs().max()) <= 1
    # angular pixels of magnitude up to ~3200 (one float32 ulp = 2.4e-4) through UT weights of ~ -99
    assert float((m2.cpu()[both] - ref["means2d"][both]).abs().max()) < 0.1
    torch.testing.assert_close(dep.cpu()[both], ref["depths"][both], rtol=1e-5, atol=1e-5)
    rel = (con.cpu()[both] - ref["conics"][both]).abs() / (ref["conics"][both].abs().max(-1, keepdim=True).values + 1e-12)
    assert float(rel.max()) < 5e-2 and float(rel.median()) < 1e-3
    rays = _ops.lidar_element_rays(t("viewmats"), vm1, lidar, rs).cpu()
    torch.testing.assert_close(rays, torch.from_numpy(gold[f"{name}.rays_{tag}"]), rtol=0, atol=5e-6)


def test_lidar_rasterization_renders_elements(G):
    """rasterization(camera_model='lidar', with_ut=True, with_eval3d=True): angle-space projection -> lidar tiling -> the
    from-world kernels on virtual pixel tiles -> [n_rows, n_columns] elements. Every element lands in exactly one tile; the
    render equals compositing each element's own ray through a ONE-tile-per-element pinhole-free path: the same op fed the
    element rays directly as an image of [n_rows, n_columns] pixels with square tiles (no lidar tiling involved)."""
    from gsplat_amd import _ops

    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "lidar_ref.npz")))
    name = "cw_120"
    lidar = lidar_from_golden(gold, name)
    t = lambda k: torch.from_numpy(gold[f"{name}.cam.{k}"]).to(DEV)  # noqa: E731
    n_rows, n_cols = int(lidar.row_elevations_rad.shape[0]), int(lidar.column_azimuths_rad.shape[0])
    pack = lidar.tiles_pack_info.cpu()
    emap = lidar.tiles_to_elements_map.cpu()
    assert int(pack[:, 1].sum()) == n_rows * n_cols and emap.shape[0] == n_rows * n_cols
    assert torch.unique(emap[:, 1].long() * n_cols + emap[:, 0].long()).numel() == n_rows * n_cols
    leaves = {k: t(k).clone().requires_grad_(True) for k in ("pts", "quats", "scales", "opac")}
    colors = torch.rand(leaves["pts"].shape[0], 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    colors.requires_grad_(True)
    bg = torch.rand(2, 3, device=DEV)
    rc, ra, meta = G.rasterization(leaves["pts"], leaves["quats"], leaves["scales"] * 3.0, leaves["opac"], colors, t("viewmats"),
                                   t("Ks"), n_cols, n_rows, camera_model="lidar", lidar_coeffs=lidar, with_ut=True,
                                   with_eval3d=True, packed=False, backgrounds=bg)
    assert rc.shape == (2, n_rows, n_cols, 3) and ra.shape == (2, n_rows, n_cols, 1)
    assert bool(torch.isfinite(rc).all()) and float(ra.max()) > 0.3 and float((ra > 0).float().mean()) > 0.05
    (rc.sum() + ra.sum()).backward()
    for k, leaf in leaves.items():
        assert leaf.grad is not None and bool(torch.isfinite(leaf.grad).all()) and float(leaf.grad.abs().max()) > 0, k
    # the same Gaussians, every Gaussian offered to EVERY pixel tile (lists = all visible rows in depth order): an upper bound of
    # what any tiling can blend. Where the lidar tiling's lists are complete the two agree; a tile list can only miss Gaussians
    # whose box does not reach the tile, i.e. whose contribution is below the 3.33-sigma cut of the extent
    rays = _ops.lidar_element_rays(t("viewmats"), None, lidar, 4)
    radii, depths = meta["radii"], meta["depths"]
    vis = (radii > 0).all(-1)
    C, N = vis.shape
    ts, tw, th = 8, (n_cols + 7) // 8, (n_rows + 7) // 8
    lists, offsets, base = [], [], 0
    for c in range(C):
        rows = torch.nonzero(vis[c])[:, 0]
        rows = rows[torch.argsort(depths[c][rows], stable=True)] + c * N
        for _ in range(th * tw):
            offsets.append(base)
            lists.append(rows)
            base += rows.numel()
    fl = torch.cat(lists).to(torch.int32)
    off = torch.tensor(offsets, device=DEV, dtype=torch.int32).reshape(C, th, tw)
    cols_cn = colors.detach()[None].expand(C, N, 3).contiguous()
    full, fa, _, _, _ = G.rasterize_to_pixels_eval3d_extra(
        leaves["pts"].detach(), leaves["quats"].detach(), leaves["scales"].detach() * 3.0, cols_cn,
        meta["opacities"].detach().contiguous(), t("viewmats"), t("Ks"), n_cols, n_rows, ts, off, fl, backgrounds=bg, rays=rays)
    diff = (full - rc.detach()).abs()
    assert float(diff.mean()) < 2e-3 and float((diff > 2e-2).float().mean()) < 5e-3, (float(diff.mean()), float(diff.max()))


def test_lidar_without_eval3d_composites_the_lidar_lists_as_pixel_tiles(G):
    """rasterization(camera_model='lidar', with_ut=True, with_eval3d=False) is what the reference's orchestrator does with it
    (Rendering.cpp:1309-1425): angle-space projection -> lidar tiling -> the CLASSIC compositing kernels fed those lists as
    tile_size x tile_size pixel tiles of a [n_rows, n_columns] image. The same call chain made by hand gives the same image, the
    absgrad switch does not change the forward, and float64 boxes / depths give the lists of their float32 values (the
    reference's second instantiation narrows on load, IntersectTileLidar.cu:185-186, 387-392)."""
    from gsplat_amd import _ops  # noqa: F401

    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "lidar_ref.npz")))
    name = "cw_120"
    lidar = lidar_from_golden(gold, name)
    t = lambda k: torch.from_numpy(gold[f"{name}.cam.{k}"]).to(DEV)  # noqa: E731
    n_rows, n_cols = int(lidar.row_elevations_rad.shape[0]), int(lidar.column_azimuths_rad.shape[0])
    colors = torch.rand(t("pts").shape[0], 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    args = (t("pts"), t("quats"), t("scales") * 3.0, t("opac"), colors, t("viewmats"), t("Ks"), n_cols, n_rows)
    kw = dict(camera_model="lidar", lidar_coeffs=lidar, with_ut=True, with_eval3d=False, packed=False, tile_size=16,
              render_mode="RGB+D")
    rc, ra, meta = G.rasterization(*args, **kw)
    assert rc.shape == (2, n_rows, n_cols, 4) and bool(torch.isfinite(rc).all()) and bool(torch.isfinite(ra).all())
    assert meta["tile_width"] == int(lidar.n_bins_azimuth) and meta["tile_height"] == int(lidar.n_bins_elevation)
    rc2, ra2, _ = G.rasterization(*args, absgrad=True, **kw)
    assert torch.equal(rc, rc2) and torch.equal(ra, ra2)
    feats = torch.cat([colors[None].expand(2, -1, -1), meta["depths"][..., None]], -1).contiguous()
    hc, ha = G.rasterize_to_pixels(meta["means2d"], meta["conics"], feats, meta["opacities"], n_cols, n_rows, 16,
                                   meta["isect_offsets"], meta["flatten_ids"])
    assert torch.equal(hc, rc) and torch.equal(ha, ra)
    # float64 instantiation of the lidar tiling
    m2, rad, dep = meta["means2d"], meta["radii"], meta["depths"]
    a32 = torch.ops.gsplat.intersect_tile_lidar(lidar, m2, rad, dep, None, None, None, True, False)
    a64 = torch.ops.gsplat.intersect_tile_lidar(lidar, m2.double(), rad, dep.double(), None, None, None, True, False)
    assert all(torch.equal(x, y) for x, y in zip(a32, a64))
