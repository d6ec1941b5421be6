"""GPU parity of the sparse pixel-set path: layout builder and masked intersection
against the oracle (exact), sparse compositing / query rasterizers against the dense kernels at the requested pixels
(forward exact: same arithmetic on the same per-tile lists; gradients scale-relative, atomics reorder sums), and against
the CPU oracle's dense rasterization directly."""
import math

import numpy as np
import pytest
import torch

from _util import assert_close_ratio, assert_grad_close, make_scene

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def G():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    import gsplat_amd

    return gsplat_amd


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    return oracle


def cpu(t):
    return None if t is None else t.detach().cpu()


def _pixels(P, I, W, H, seed, clustered=False):
    g = torch.Generator().manual_seed(seed)
    if clustered:  # a few dense patches + scattered pixels: full tiles, partial tiles, untouched tiles
        sel = torch.zeros(I, H, W, dtype=torch.bool)
        sel[0, 3:29, 10:45] = True
        sel[I - 1, H - 9:, W - 21:] = True
        sel.view(-1)[torch.randperm(I * H * W, generator=g)[:P]] = True
        flat = torch.nonzero(sel.view(-1))[:, 0]
        flat = flat[torch.randperm(flat.numel(), generator=g)]  # caller order is arbitrary
    else:
        flat = torch.randperm(I * H * W, generator=g)[:P]
    img, rem = flat // (H * W), flat % (H * W)
    return torch.stack([rem // W, rem % W], -1).to(torch.int32).to(DEV), img.to(torch.int32).to(DEV)


def _scene(G, N, C, W, H, seed, packed=False):
    sc, W, H = make_scene(N=N, C=C, width=W, height=H, seed=seed)
    a = {k: v.to(DEV) for k, v in sc.items()}
    rad, m2, d, con, _ = G.fully_fused_projection(a["means"], None, a["quats"], a["scales"], a["viewmats"], a["Ks"], W, H,
                                                  opacities=a["opacities"])
    op = a["opacities"][None].expand(C, -1).contiguous()
    ci = None
    if packed:
        vis = (rad > 0).all(-1)
        ci, _gi = torch.where(vis)
        rad, m2, d, con, op = rad[vis], m2[vis], d[vis], con[vis], op[vis]
    return rad, m2, d, con, op, ci


@pytest.mark.parametrize("tile_size", [16, 8])
def test_layout_on_device_matches_oracle(G, O, tile_size):
    I, W, H = 3, 83, 61
    pixels, image_ids = _pixels(1500, I, W, H, seed=5, clustered=True)
    tw, th = math.ceil(W / tile_size), math.ceil(H / tile_size)
    out = G.build_sparse_tile_layout(pixels, image_ids, I, tile_size, tw, th)
    ref = O.sparse_tile_layout(cpu(pixels), cpu(image_ids), I, tile_size, tw, th)
    assert out[2].dtype == torch.uint64 and out[0].dtype == torch.int32
    for got, want in zip(out, ref):
        got = cpu(got)
        got = got.view(torch.int64).numpy().view(np.uint64) if got.dtype == torch.uint64 else got.numpy()
        assert np.array_equal(got, want)


@pytest.mark.parametrize("path", [None, "binned", "legacy"])
@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("C", [1, 2])
def test_isect_sparse_exact(G, O, packed, C, path, monkeypatch):
    # C == 1 packed and every dense case run the fused masked kernels; C == 2 packed takes the enumerate-and-filter route.
    # `path`: the tile-owner-major intersection is chosen on its own only for 200 k - 1.2 M rows per 1080p image; forced here so
    # that the sparse caller of both fused paths is exercised at test size (the binned branch used to crash)
    if path is not None:
        monkeypatch.setenv("GSX_ISECT_PATH", path)
    W, H, ts = 150, 100, 16
    rad, m2, d, con, op, ci = _scene(G, 3000, C, W, H, seed=17, packed=packed)
    tw, th = math.ceil(W / ts), math.ceil(H / ts)
    pixels, image_ids = _pixels(700, C, W, H, seed=3, clustered=True)
    act, tmask, _pm, _cum, _map = G.build_sparse_tile_layout(pixels, image_ids, C, ts, tw, th)
    off, fl = G.isect_tiles_sparse(m2, rad, d, tmask, act, C, ts, tw, th, image_ids=ci)
    off_o, fl_o = O.isect_tiles_sparse(cpu(m2), cpu(rad), cpu(d), cpu(tmask), cpu(act), C, ts, tw, th, image_ids=cpu(ci))
    assert off.dtype == torch.int32 and off.shape == (act.numel() + 1,) and fl.dtype == torch.int32
    assert torch.equal(cpu(off), off_o)
    assert torch.equal(cpu(fl), fl_o)
    assert int(off[-1]) == fl.numel() > 0


def test_isect_sparse_edge_cases(G):
    ts, tw, th = 16, 4, 3
    m2 = torch.rand(1, 50, 2, device=DEV) * 40
    rad = torch.full((1, 50, 2), 5, dtype=torch.int32, device=DEV)
    d = torch.rand(1, 50, device=DEV)
    none = torch.zeros(1, th, tw, dtype=torch.bool, device=DEV)
    off, fl = G.isect_tiles_sparse(m2, rad, d, none, torch.zeros(0, dtype=torch.int32, device=DEV), 1, ts, tw, th)
    assert off.tolist() == [0] and fl.numel() == 0
    # an active tile that nothing reaches: empty range, sentinel still n_isects
    far = torch.zeros(1, th, tw, dtype=torch.bool, device=DEV)
    far[0, 2, 3] = True
    far[0, 0, 0] = True
    act = torch.nonzero(far.view(-1))[:, 0].to(torch.int32)
    off, fl = G.isect_tiles_sparse(m2, rad, d, far, act, 1, ts, tw, th)
    assert off.shape == (3,) and off[1] == off[2] == fl.numel() and off[0] == 0 and fl.numel() > 0
    # all radii zero
    off, fl = G.isect_tiles_sparse(m2, torch.zeros_like(rad), d, far, act, 1, ts, tw, th)
    assert off.tolist() == [0, 0, 0] and fl.numel() == 0


@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_layout_and_isect_sparse_vs_reference_golden(G, case):
    """Outputs of the reference's own looped torch references (_torch_impl.py:485-708), tests/golden/sparse_ref.npz."""
    import os

    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sparse_ref.npz")))
    C, N, W, H, ts, P = g[f"{case}_dims"].tolist()
    tw, th = math.ceil(W / ts), math.ceil(H / ts)
    t = lambda k: torch.from_numpy(g[f"{case}_{k}"]).to(DEV)  # noqa: E731
    out = G.build_sparse_tile_layout(t("pixels"), t("image_ids"), C, ts, tw, th)
    for got, k in zip(out, ("active_tiles", "tile_mask", "pixel_mask", "pixel_cumsum", "pixel_map")):
        got = cpu(got)
        got = got.view(torch.int64).numpy().view(np.uint64) if got.dtype == torch.uint64 else got.numpy()
        assert np.array_equal(got, g[f"{case}_{k}"]), k
    m2, rad, d = t("means2d"), t("radii"), t("depths")
    off, fl = G.isect_tiles_sparse(m2, rad, d, out[1], out[0], C, ts, tw, th)
    assert np.array_equal(cpu(off).numpy(), g[f"{case}_tile_offsets"])
    assert np.array_equal(cpu(fl).numpy(), g[f"{case}_flatten_ids"])
    vis = (rad > 0).all(-1)
    ci, _ = torch.where(vis)
    off, fl = G.isect_tiles_sparse(m2[vis], rad[vis], d[vis], out[1], out[0], C, ts, tw, th, image_ids=ci.to(torch.int32))
    assert np.array_equal(cpu(off).numpy(), g[f"{case}_tile_offsets_packed"])
    assert np.array_equal(cpu(fl).numpy(), g[f"{case}_flatten_ids_packed"])


def _sparse_vs_dense(G, O, N, C, W, H, ts, D, seed, packed=False, bg=False, masks=False, absgrad=False, clustered=True,
                     P=900):
    rad, m2, d, con, op, ci = _scene(G, N, C, W, H, seed=seed, packed=packed)
    tw, th = math.ceil(W / ts), math.ceil(H / ts)
    g = torch.Generator().manual_seed(seed)
    colors = torch.rand(op.shape + (D,), generator=g).to(DEV)
    backgrounds = torch.rand(C, D, generator=g).to(DEV) if bg else None
    tile_masks = (torch.rand(C, th, tw, generator=g) > 0.3).to(DEV) if masks else None
    pixels, image_ids = _pixels(P, C, W, H, seed=seed + 1, clustered=clustered)
    P = pixels.shape[0]
    act, tmask, pmask, cum, pmap = G.build_sparse_tile_layout(pixels, image_ids, C, ts, tw, th)
    off_s, fl_s = G.isect_tiles_sparse(m2, rad, d, tmask, act, C, ts, tw, th, image_ids=ci)
    # dense reference run on the AABB intersection lists (the sparse enumeration is AABB-only)
    if packed:
        _, ids, fl = G.isect_tiles(m2, rad, d, ts, tw, th, packed=True, n_images=C, image_ids=ci,
                                   gaussian_ids=torch.zeros_like(ci))
    else:
        _, ids, fl = G.isect_tiles(m2, rad, d, ts, tw, th)
    off = G.isect_offset_encode(ids, C, tw, th)
    sel = (image_ids.long(), pixels[:, 0].long(), pixels[:, 1].long())

    def leaves():
        ls = [t.clone().requires_grad_(True) for t in (m2, con, colors, op)]
        return ls, (backgrounds.clone().requires_grad_(True) if bg else None)

    ls_s, bg_s = leaves()
    rc_s, ra_s = G.rasterize_to_pixels_sparse(ls_s[0], ls_s[1], ls_s[2], ls_s[3], image_ids, act, off_s, fl_s, pmask, cum,
                                              pmap, W, H, ts, tw, th, backgrounds=bg_s, masks=tile_masks, packed=packed,
                                              absgrad=absgrad)
    ls_d, bg_d = leaves()
    rc_d, ra_d = G.rasterize_to_pixels(ls_d[0], ls_d[1], ls_d[2], ls_d[3], W, H, ts, off, fl, backgrounds=bg_d,
                                       masks=tile_masks, packed=packed, absgrad=absgrad)
    assert rc_s.shape == (P, D) and ra_s.shape == (P, 1)
    assert torch.equal(rc_s, rc_d[sel]), "sparse colours differ from the dense render at the requested pixels"
    assert torch.equal(ra_s, ra_d[sel])
    # and against the CPU oracle's dense rasterization
    rc_o, ra_o, _ = O.rasterize_to_pixels(cpu(m2), cpu(con), cpu(colors), cpu(op), W, H, ts, cpu(off), cpu(fl),
                                          backgrounds=cpu(backgrounds), masks=cpu(tile_masks))
    sel_c = tuple(s.cpu() for s in sel)
    assert_close_ratio(cpu(rc_s), rc_o[sel_c], 1e-4, 2e-5, max_bad_ratio=2e-4, name="sparse colours vs oracle")
    assert_close_ratio(cpu(ra_s), ra_o[sel_c], 1e-4, 2e-5, max_bad_ratio=2e-4, name="sparse alphas vs oracle")

    v_rc, v_ra = torch.randn(P, D, generator=g).to(DEV), torch.randn(P, 1, generator=g).to(DEV)
    ((rc_s * v_rc).sum() + (ra_s * v_ra).sum()).backward()
    ((rc_d[sel] * v_rc).sum() + (ra_d[sel] * v_ra).sum()).backward()
    for a, b, name in zip(ls_s, ls_d, ("v_means2d", "v_conics", "v_colors", "v_opacities")):
        assert a.grad is not None and a.grad.abs().sum() > 0, name
        assert_grad_close(cpu(a.grad), cpu(b.grad), rel=2e-4, max_bad_ratio=1e-5, name=name)
    if bg:
        assert_grad_close(cpu(bg_s.grad), cpu(bg_d.grad), rel=1e-4, name="v_backgrounds")
    if absgrad:
        assert_grad_close(cpu(ls_s[0].absgrad), cpu(ls_d[0].absgrad), rel=2e-4, max_bad_ratio=1e-5, name="absgrad")


@pytest.mark.parametrize("D", [3, 1, 40, 20, 9])
def test_sparse_raster_channels(G, O, D):
    """D = 40 / 20: the matrix-core forward (from 17 channels per launch: bit-equal to the dense render because its groups are
    fixed list positions) and backward (5 .. 32 channels) on a sparse pixel set; D = 9: the four-wave forward with that backward."""
    _sparse_vs_dense(G, O, N=2500, C=2, W=150, H=100, ts=16, D=D, seed=21, bg=(D in (3, 20)))


@pytest.mark.parametrize("ts", [8, 4])
def test_sparse_raster_tile_sizes(G, O, ts):
    _sparse_vs_dense(G, O, N=1500, C=1, W=90, H=70, ts=ts, D=3, seed=22)


def test_sparse_raster_packed_masks_absgrad(G, O):
    _sparse_vs_dense(G, O, N=2500, C=2, W=150, H=100, ts=16, D=3, seed=23, packed=True, bg=True, masks=True, absgrad=True)
    _sparse_vs_dense(G, O, N=2500, C=1, W=150, H=100, ts=16, D=3, seed=24, packed=True, bg=True)


def test_sparse_raster_every_pixel_and_single_pixel(G, O):
    _sparse_vs_dense(G, O, N=1500, C=1, W=64, H=48, ts=16, D=3, seed=25, clustered=False, P=64 * 48, bg=True)
    _sparse_vs_dense(G, O, N=1500, C=2, W=64, H=48, ts=16, D=3, seed=26, clustered=False, P=1)


def test_sparse_raster_empty_pixel_set(G):
    rad, m2, d, con, op, _ = _scene(G, 500, 1, 64, 48, seed=27)
    colors = torch.rand(1, 500, 3, device=DEV)
    pixels, image_ids = torch.zeros((0, 2), dtype=torch.int32, device=DEV), torch.zeros(0, dtype=torch.int32, device=DEV)
    act, tmask, pmask, cum, pmap = G.build_sparse_tile_layout(pixels, image_ids, 1, 16, 4, 3)
    off, fl = G.isect_tiles_sparse(m2, rad, d, tmask, act, 1, 16, 4, 3)
    rc, ra = G.rasterize_to_pixels_sparse(m2, con, colors, op, image_ids, act, off, fl, pmask, cum, pmap, 64, 48, 16, 4, 3)
    assert rc.shape == (0, 3) and ra.shape == (0, 1)


@pytest.mark.parametrize("packed", [False, True])
def test_sparse_query_rasterizers_match_dense(G, packed):
    C, W, H, ts = 2, 72, 56, 16
    rad, m2, d, con, op, ci = _scene(G, 1500, C, W, H, seed=31, packed=packed)
    tw, th = math.ceil(W / ts), math.ceil(H / ts)
    pixels, image_ids = _pixels(600, C, W, H, seed=8, clustered=True)
    P = pixels.shape[0]
    act, tmask, pmask, cum, pmap = G.build_sparse_tile_layout(pixels, image_ids, C, ts, tw, th)
    off_s, fl_s = G.isect_tiles_sparse(m2, rad, d, tmask, act, C, ts, tw, th, image_ids=ci)
    if packed:
        _, ids, fl = G.isect_tiles(m2, rad, d, ts, tw, th, packed=True, n_images=C, image_ids=ci,
                                   gaussian_ids=torch.zeros_like(ci))
    else:
        _, ids, fl = G.isect_tiles(m2, rad, d, ts, tw, th)
    off = G.isect_offset_encode(ids, C, tw, th)
    sel = (image_ids.long(), pixels[:, 0].long(), pixels[:, 1].long())
    layout = (act, off_s, fl_s, pmask, cum, pmap)

    cnt_s, al_s = G.rasterize_num_contributing_gaussians_sparse(m2, con, op, *layout, W, H, ts, tw, th)
    cnt_d, al_d = G.rasterize_num_contributing_gaussians(m2, con, op, off, fl, W, H, ts)
    assert cnt_s.shape == (P,) and cnt_s.dtype == torch.int32 and int(cnt_s.max()) > 3
    assert torch.equal(cnt_s, cnt_d[sel]) and torch.equal(al_s, al_d[sel])

    ids_s, w_s = G.rasterize_contributing_gaussian_ids_sparse(m2, con, op, *layout, cnt_s, W, H, ts, tw, th)
    ids_d, w_d = G.rasterize_contributing_gaussian_ids(m2, con, op, off, fl, W, H, ts, cnt_d)
    K = int(cnt_s.max())
    assert ids_s.shape == (P, K)
    assert torch.equal(ids_s, ids_d[sel][:, :K]) and torch.equal(w_s, w_d[sel][:, :K])

    tid_s, tw_s = G.rasterize_top_contributing_gaussian_ids_sparse(m2, con, op, *layout, W, H, ts, tw, th, 4)
    tid_d, tw_d = G.rasterize_top_contributing_gaussian_ids(m2, con, op, off, fl, W, H, ts, 4)
    assert tid_s.shape == (P, 4)
    assert torch.equal(tid_s, tid_d[sel]) and torch.equal(tw_s, tw_d[sel])
