"""GPU parity of the 2DGS path (projection_2dgs_*, rasterize_to_pixels_2dgs, rasterization_2dgs) against the CPU
oracle on the same seeded inputs and against the committed outputs of the reference's Python (tests/golden/
garden_quarter_2dgs.npz). Tolerances: the reference's own for this path (tests/test_2dgs.py: rtol/atol 1e-4 forward,
scale-relative gradients)."""
import math
import os

import numpy as np
import pytest
import torch

from _util import assert_close_ratio, assert_grad_close, make_scene, to_t

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


@pytest.fixture(scope="module")
def g2():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "garden_quarter_2dgs.npz")))


def cpu(t):
    return None if t is None else t.detach().cpu()


def _proj_compare(G, O, sc, W, H, pose):
    names = ("means", "quats", "scales", "viewmats")
    lg = [sc[k].to(DEV).clone().requires_grad_(k != "viewmats" or pose) for k in names]
    lo = [sc[k].clone().requires_grad_(k != "viewmats" or pose) for k in names]
    rad, m2, d, M, n = G.fully_fused_projection_2dgs(*lg, sc["Ks"].to(DEV), W, H)
    rad_o, m2_o, d_o, M_o, n_o = O.fully_fused_projection_2dgs(*lo, sc["Ks"], W, H)
    vis_g, vis_o = (cpu(rad) > 0).all(-1), (rad_o > 0).all(-1)
    assert (vis_g == vis_o).float().mean() > 0.999
    valid = vis_g & vis_o
    assert (cpu(rad)[valid] - rad_o[valid]).abs().max() <= 1
    assert_close_ratio(cpu(m2)[valid], m2_o[valid], 1e-4, 1e-3, max_bad_ratio=5e-4, name="means2d")
    assert_close_ratio(cpu(d)[valid], d_o[valid], 1e-5, 1e-5, name="depths")
    assert_close_ratio(cpu(M)[valid], M_o[valid], 1e-4, 1e-3, name="ray_transforms")
    assert_close_ratio(cpu(n)[valid], n_o[valid], 1e-5, 1e-5, name="normals")
    assert (cpu(m2)[~vis_g] == 0).all()  # culled rows are zero-filled
    g = torch.Generator().manual_seed(1)
    w = [torch.randn(t.shape, generator=g) for t in (m2_o, d_o, M_o, n_o)]
    vm = valid.float()

    def loss(m2_, d_, M_, n_, dev):
        ws = [x.to(dev) for x in w]
        v = vm.to(dev)
        return ((m2_ * ws[0]).sum(-1) * v).sum() + (d_ * ws[1] * v).sum() + ((M_ * ws[2]).sum((-1, -2)) * v).sum() \
            + ((n_ * ws[3]).sum(-1) * v).sum()

    loss(m2, d, M, n, DEV).backward()
    loss(m2_o, d_o, M_o, n_o, "cpu").backward()
    for nm, a, b in zip(names, lg, lo):
        if b.grad is None:
            continue
        ga, gb = cpu(a.grad), b.grad
        if nm == "scales":
            assert (ga[..., 2] == 0).all()
            ga, gb = ga[..., :2], gb[..., :2]
        assert_grad_close(ga, gb, rel=2e-3, name="v_" + nm)


@pytest.mark.parametrize("pose", [False, True])
def test_projection_2dgs_dense_fwd_bwd(G, O, pose):
    sc, W, H = make_scene(N=4000, C=3, width=200, height=136, seed=3)
    _proj_compare(G, O, sc, W, H, pose)


@pytest.mark.parametrize("sparse_grad,C", [(False, 2), (True, 2), (True, 1)])
def test_projection_2dgs_packed_matches_dense(G, sparse_grad, C):
    sc, W, H = make_scene(N=3000, C=C, width=160, height=120, seed=4)
    dsc = {k: v.to(DEV) for k, v in sc.items()}
    names = ("means", "quats", "scales", "viewmats")
    ld = [dsc[k].clone().requires_grad_(True) for k in names]
    lp = [dsc[k].clone().requires_grad_(True) for k in names]
    rad, m2, d, M, n = G.fully_fused_projection_2dgs(*ld, dsc["Ks"], W, H)
    bi, ci, gi, indptr, rad_p, m2_p, d_p, M_p, n_p = G.fully_fused_projection_2dgs(*lp, dsc["Ks"], W, H, packed=True,
                                                                                   sparse_grad=sparse_grad)
    vis = (rad > 0).all(-1)
    c_ref, g_ref = torch.where(vis)
    assert torch.equal(ci, c_ref) and torch.equal(gi, g_ref) and (bi == 0).all()
    assert torch.equal(indptr.long(), torch.cat([torch.zeros(1, device=DEV, dtype=torch.long), vis.sum(-1).cumsum(0)]))
    for a, b in ((rad_p, rad[vis]), (m2_p, m2[vis]), (d_p, d[vis]), (M_p, M[vis]), (n_p, n[vis])):
        assert torch.equal(a, b)
    g = torch.Generator().manual_seed(2)
    w = [torch.randn(t.shape, generator=g).to(DEV) for t in (m2_p, d_p, M_p, n_p)]
    ((m2_p * w[0]).sum() + (d_p * w[1]).sum() + (M_p * w[2]).sum() + (n_p * w[3]).sum()).backward()
    ((m2[vis] * w[0]).sum() + (d[vis] * w[1]).sum() + (M[vis] * w[2]).sum() + (n[vis] * w[3]).sum()).backward()
    for nm, a, b in zip(names, lp, ld):
        ga = a.grad
        if sparse_grad and nm != "viewmats":  # reference layout: Projection.cpp:1780-1863
            assert ga.is_sparse and ga._nnz() == gi.numel(), nm
            assert torch.equal(ga._indices(), gi[None]), nm
            ga = ga.to_dense()
        assert_grad_close(cpu(ga), cpu(b.grad), rel=1e-4, name="packed v_" + nm)


def _raster2d_case(G, O, N, C, W, H, tile_size, D, seed, bg=False, masks=False, absgrad=False, distloss=True,
                   scale_range=(0.02, 0.15)):
    sc, W, H = make_scene(N=N, C=C, width=W, height=H, seed=seed, scale_range=scale_range)
    dsc = {k: v.to(DEV) for k, v in sc.items()}
    rad, m2, d, M, nrm = G.fully_fused_projection_2dgs(dsc["means"], dsc["quats"], dsc["scales"], dsc["viewmats"],
                                                       dsc["Ks"], W, H)
    op = dsc["opacities"][None].expand(C, -1).contiguous()
    tw, th = math.ceil(W / tile_size), math.ceil(H / tile_size)
    _, ids, fl = G.isect_tiles(m2, rad, d, tile_size, tw, th)
    off = G.isect_offset_encode(ids, C, tw, th)
    g = torch.Generator().manual_seed(seed)
    colors = torch.cat([torch.rand(C, N, D - 1, generator=g).to(DEV), d[..., None]], -1).contiguous()
    backgrounds = torch.rand(C, D, generator=g).to(DEV) if bg else None
    tile_masks = (torch.rand(C, th, tw, generator=g) > 0.3).to(DEV) if masks else None
    leaves = [t.clone().requires_grad_(True) for t in (m2, M, colors, op, nrm)]
    densify = torch.zeros_like(m2).requires_grad_(True)
    bgl = backgrounds.clone().requires_grad_(True) if bg else None
    outs = G.rasterize_to_pixels_2dgs(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], densify, W, H, tile_size,
                                      off, fl, backgrounds=bgl, masks=tile_masks, absgrad=absgrad, distloss=distloss)
    ref = O.rasterize_to_pixels_2dgs(cpu(m2), cpu(M), cpu(colors), cpu(op), cpu(nrm), W, H, tile_size, cpu(off), cpu(fl),
                                     backgrounds=cpu(backgrounds), masks=cpu(tile_masks), distloss=distloss)
    names = ("render_colors", "render_alphas", "render_normals", "render_distort", "render_median")
    for nm, a, b in zip(names, outs, ref[:5]):
        # median depth flips between neighbouring surfels when T crosses 0.5 within rounding: allow a few more
        ratio = 2e-3 if nm in ("render_median", "render_distort") else 2e-4
        # ... and no cap on HOW far such a flipped pixel is off: it carries another surfel's depth (a step, not an error)
        assert_close_ratio(cpu(a), b, 2e-4, 5e-5, max_bad_ratio=ratio, name=nm,
                           **({"outlier_cap": None} if nm == "render_median" else {}))
    v = [torch.randn(b.shape, generator=g) for b in ref[:5]]
    if not distloss:
        v[3].zero_()
    sum((a * b.to(DEV)).sum() for a, b in zip(outs, v)).backward()
    go = O.rasterize_to_pixels_2dgs_bwd(cpu(m2), cpu(M), cpu(colors), cpu(op), cpu(nrm), W, H, tile_size, cpu(off),
                                        cpu(fl), ref[0], ref[1], ref[5], ref[6], v[0], v[1], v[2],
                                        v[3] if distloss else None, v[4], backgrounds=cpu(backgrounds),
                                        masks=cpu(tile_masks), absgrad=absgrad)
    for leaf, key in zip(leaves + [densify], ("v_means2d", "v_ray_transforms", "v_colors", "v_opacities", "v_normals",
                                              "v_densify")):
        assert_grad_close(cpu(leaf.grad), go[key].reshape(leaf.shape), rel=3e-3, max_bad_ratio=5e-4, name=key)
    if bg:
        assert_grad_close(cpu(bgl.grad), go["v_backgrounds"], rel=1e-3, name="v_backgrounds")
    if absgrad:
        assert_grad_close(cpu(leaves[0].absgrad), go["v_means2d_abs"].reshape(leaves[0].shape), rel=3e-3,
                          max_bad_ratio=5e-4, name="absgrad")


@pytest.mark.parametrize("D", [1, 4, 7, 20])
def test_rasterize_2dgs_channels(G, O, D):
    _raster2d_case(G, O, N=5000, C=2, W=200, H=136, tile_size=16, D=D, seed=50 + D, bg=True)


@pytest.mark.parametrize("tile_size", [16, 8])
def test_rasterize_2dgs_tile_sizes_masks_absgrad(G, O, tile_size):
    _raster2d_case(G, O, N=3000, C=1, W=150, H=100, tile_size=tile_size, D=4, seed=60 + tile_size, bg=True, masks=True,
                   absgrad=True)


def test_rasterize_2dgs_no_distloss_long_lists(G, O):
    _raster2d_case(G, O, N=6000, C=1, W=96, H=64, tile_size=16, D=4, seed=70, distloss=False, scale_range=(0.2, 0.6))


def test_golden_2dgs_vs_reference_outputs(G, g2):
    """HIP kernels against outputs of the REFERENCE's _torch_impl_2dgs.py (+ autograd), committed fixture."""
    W, H = (int(v) for v in g2["wh"])
    leaves = [to_t(g2[k], DEV).clone().requires_grad_(True) for k in ("means", "quats", "scales", "viewmats")]
    rad, m2, d, M, n = G.fully_fused_projection_2dgs(*leaves, to_t(g2["Ks"], DEV), W, H)
    valid = to_t(g2["proj_valid"])
    assert ((cpu(rad) > 0).all(-1) == (to_t(g2["proj_radii"]) > 0).all(-1)).float().mean() > 0.999
    # mean2d = sum(f M0 M2) with f = 1/distance: surfels seen almost edge-on (distance -> 0) are ill-conditioned in
    # fp32, a handful of rows differ between any two evaluation orders
    assert_close_ratio(cpu(m2)[valid], to_t(g2["proj_means2d"])[valid], 1e-4, 1e-3, max_bad_ratio=5e-4, name="means2d")
    assert_close_ratio(cpu(M)[valid], to_t(g2["proj_ray_transforms"])[valid], 1e-4, 1e-3, name="ray_transforms")
    assert_close_ratio(cpu(n)[valid], to_t(g2["proj_normals"])[valid], 1e-5, 1e-5, name="normals")
    vm = valid.float().to(DEV)
    loss = ((m2 * to_t(g2["proj_w_means2d"], DEV)).sum(-1) * vm).sum() + (d * to_t(g2["proj_w_depths"], DEV) * vm).sum() \
        + ((M * to_t(g2["proj_w_ray_transforms"], DEV)).sum((-1, -2)) * vm).sum() \
        + ((n * to_t(g2["proj_w_normals"], DEV)).sum(-1) * vm).sum()
    loss.backward()
    for nm, leaf in zip(("v_means", "v_quats", "v_scales", "v_viewmats"), leaves):
        a, e = cpu(leaf.grad), to_t(g2["proj_" + nm])
        if nm == "v_scales":
            a, e = a[:, :2], e[:, :2]
        assert_grad_close(a, e, rel=2e-3, name=nm)
    Wr, Hr, ts = (int(v) for v in g2["rast_wh"])
    t = {k: to_t(g2["rast_" + k], DEV) for k in ("means2d", "ray_transforms", "colors", "opacities", "normals",
                                                 "offsets", "flatten_ids", "backgrounds")}
    lv = [t[k].clone().requires_grad_(True) for k in ("means2d", "ray_transforms", "colors", "opacities", "normals",
                                                      "backgrounds")]
    densify = torch.zeros_like(lv[0]).requires_grad_(True)
    rc, ra, rn, rd, rm = G.rasterize_to_pixels_2dgs(lv[0], lv[1], lv[2], lv[3], lv[4], densify, Wr, Hr, ts,
                                                    t["offsets"], t["flatten_ids"], backgrounds=lv[5])
    assert_close_ratio(cpu(rc), g2["rast_render_colors"], 2e-4, 5e-5, max_bad_ratio=2e-4, name="render_colors")
    assert_close_ratio(cpu(ra), g2["rast_render_alphas"], 2e-4, 5e-5, max_bad_ratio=2e-4, name="render_alphas")
    assert_close_ratio(cpu(rn), g2["rast_render_normals"], 2e-4, 5e-5, max_bad_ratio=2e-4, name="render_normals")
    ((rc * to_t(g2["rast_v_render_colors"], DEV)).sum() + (ra * to_t(g2["rast_v_render_alphas"], DEV)).sum()
     + (rn * to_t(g2["rast_v_render_normals"], DEV)).sum()).backward()
    for leaf, key in zip(lv, ("v_means2d", "v_ray_transforms", "v_colors", "v_opacities", "v_normals", "v_backgrounds")):
        assert_grad_close(cpu(leaf.grad), g2["rast_" + key], rel=3e-3, max_bad_ratio=5e-4, name=key)


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("render_mode,sh_degree,distloss", [("RGB", None, False), ("RGB+ED", 3, True), ("D", None, False)])
def test_rasterization_2dgs_pipeline_matches_oracle(G, O, packed, render_mode, sh_degree, distloss):
    """End-to-end rasterization_2dgs(): forward outputs + gradients to the leaves vs the oracle stages chained with
    torch autograd (projection: torch; compositing: C oracle backward injected)."""
    sc, W, H = make_scene(N=2500, C=2, width=160, height=112, seed=9, sh_degree=sh_degree)
    _pipeline_2dgs_vs_oracle(G, O, sc, W, H, packed, render_mode, sh_degree, distloss)


def test_c5_matches_oracle(G, O):
    """BASELINE.json configs[4] (c5) at full size - 1 M surfels, 1080p, SH degree 3, RGB+ED + normals + distortion - against
    the oracle chain (OpenMP C compositing + torch-CPU projection / SH; a few seconds per pass on the host cores)."""
    import bench

    O.set_threads(min(os.cpu_count() or 1, 32))
    sc, W, H = bench.make_workload(1_000_000, "cpu")
    for packed in (False, True):
        _pipeline_2dgs_vs_oracle(G, O, sc, W, H, packed, "RGB+ED", 3, True, full_size=True)


def _pipeline_2dgs_vs_oracle(G, O, sc, W, H, packed, render_mode, sh_degree, distloss, full_size=False):
    C, N = sc["viewmats"].shape[0], sc["means"].shape[0]
    names = ("means", "quats", "scales", "opacities", "colors")
    lg = {k: sc[k].to(DEV).clone().requires_grad_(True) for k in names}
    out = G.rasterization_2dgs(lg["means"], lg["quats"], lg["scales"], lg["opacities"], lg["colors"],
                               sc["viewmats"].to(DEV), sc["Ks"].to(DEV), W, H, sh_degree=sh_degree, packed=packed,
                               render_mode=render_mode, distloss=distloss)
    rc, ra, rn, sn, rd, rm, meta = out
    # oracle chain
    lo = {k: sc[k].clone().requires_grad_(True) for k in names}
    rad, m2, d, M, nrm = O.fully_fused_projection_2dgs(lo["means"], lo["quats"], lo["scales"], sc["viewmats"], sc["Ks"],
                                                       W, H)
    op = lo["opacities"][None].expand(C, -1)
    has_color, has_depth = render_mode != "D", render_mode != "RGB"
    feats = None
    if has_color:
        if sh_degree is None:
            feats = lo["colors"][None].expand(C, -1, -1)
        else:
            feats = torch.clamp_min(O.spherical_harmonics(sh_degree, lo["means"][None], sc["viewmats"][None],
                                                          lo["colors"], (rad > 0).all(-1)[None])[0] + 0.5, 0.0)
    if has_depth:
        feats = d[..., None] if feats is None else torch.cat([feats, d[..., None]], -1)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    _, ids, fl = O.isect_tiles(m2, rad, d, 16, tw, th, sort=True)
    off = O.isect_offset_encode(ids, C, tw, th)
    ref = O.rasterize_to_pixels_2dgs(m2, M, feats, op, nrm, W, H, 16, off, fl, distloss=distloss)
    rc_o, ra_o, rn_o, rd_o, rm_o, li_o, mi_o = ref
    expected = render_mode == "RGB+ED"
    a_ = ra_o.clamp_min(1e-10)
    rc_cmp = torch.cat([rc_o[..., :-1], rc_o[..., -1:] / a_], -1) if expected else rc_o
    R_c2w = torch.linalg.inv(sc["viewmats"])[:, :3, :3]
    rn_cmp = torch.einsum("cij,chwj->chwi", R_c2w, rn_o)
    # radii are ceil() of float expressions: the reference compares them with atol=1 (tests/test_basic.py), so the
    # intersection COUNT may differ by a few entries between the GPU and the torch-CPU oracle projection
    assert abs(meta["isect_ids"].numel() - ids.numel()) <= max(4, ids.numel() // 2000)
    if full_size and not packed:  # ... and nothing but those +-1 radii is behind the difference
        r_g, r_o = cpu(meta["radii"]).reshape(-1, 2), rad.reshape(-1, 2)
        vis_g, vis_o = (r_g > 0).all(-1), (r_o > 0).all(-1)
        both = vis_g & vis_o  # a surfel exactly on a culling threshold may be visible on one side only
        assert float((vis_g != vis_o).float().mean()) < 1e-3  # same bound as the projection test above (_proj_compare)
        dr = (r_g[both] - r_o[both]).abs().float()
        # The reference's extent formula is ill-conditioned at this image size: ext^2 = m^2 - t with m up to 1920 (m^2 ~
        # 3.7e6, one fp32 ulp = 0.25 .. 0.5) and ext^2 of order 1 .. 10 for a small surfel, so two fp32 evaluation orders
        # legitimately differ by one pixel in ceil(3.33 ext) on ~10 % of the surfels and by more on ~1.5 % (the radius only sizes the
        # conservative tile box). The small scenes (160 px wide) hold the +-1 of the reference's own tests; here: close on
        # average, bounded, and the images / gradients below are what is held to the usual tolerances.
        assert float(dr.mean()) < 0.3 and float(dr.max()) <= 8 and float((dr > 1).any(-1).float().mean()) < 5e-2
    assert_close_ratio(cpu(rc), rc_cmp, 1e-3, 1e-4, max_bad_ratio=1e-3, name="render_colors")
    assert_close_ratio(cpu(ra), ra_o, 1e-4, 5e-5, max_bad_ratio=1e-3, name="render_alphas")
    assert_close_ratio(cpu(rn), rn_cmp, 1e-3, 1e-4, max_bad_ratio=1e-3, name="render_normals")
    assert_close_ratio(cpu(rd), rd_o, 1e-3, 1e-4, max_bad_ratio=3e-3, name="render_distort")
    assert (sn is None) == (not (has_color and has_depth))
    g = torch.Generator().manual_seed(11)
    v_rc, v_ra, v_rn, v_rd = (torch.randn(t.shape, generator=g) for t in (rc_cmp, ra_o, rn_cmp, rd_o))
    if not distloss:
        v_rd.zero_()
    ((rc * v_rc.to(DEV)).sum() + (ra * v_ra.to(DEV)).sum() + (rn * v_rn.to(DEV)).sum() + (rd * v_rd.to(DEV)).sum()).backward()
    # oracle backward: cotangents of the raw compositing outputs
    if expected:
        v_feat = torch.cat([v_rc[..., :-1], v_rc[..., -1:] / a_], -1)
        v_alpha = v_ra - (v_rc[..., -1:] * rc_o[..., -1:] / (a_ * a_)) * (ra_o > 1e-10)
    else:
        v_feat, v_alpha = v_rc, v_ra
    v_rn_cam = torch.einsum("cij,chwi->chwj", R_c2w, v_rn)
    gr = O.rasterize_to_pixels_2dgs_bwd(m2, M, feats, op, nrm, W, H, 16, off, fl, rc_o, ra_o, li_o, mi_o, v_feat, v_alpha,
                                        v_rn_cam, v_rd if distloss else None, torch.zeros_like(ra_o))

    def t(k, like):
        return torch.from_numpy(gr[k]).to(like.dtype).reshape(like.shape)

    torch.autograd.backward([m2, M, feats, op, nrm], [t("v_means2d", m2), t("v_ray_transforms", M), t("v_colors", feats),
                                                      t("v_opacities", op), t("v_normals", nrm)])
    for k in names:
        if lo[k].grad is None:
            assert lg[k].grad is None or float(lg[k].grad.abs().max()) == 0.0
            continue
        ga, gb = cpu(lg[k].grad), lo[k].grad
        if k == "scales":
            ga, gb = ga[..., :2], gb[..., :2]
        assert_grad_close(ga, gb, rel=5e-3, max_bad_ratio=1e-3, name=f"v_{k} packed={packed} mode={render_mode}")


def test_c5_full_size_properties_1m_surfels_1080p(G):
    """BASELINE.json configs[4] (c5): 1M surfels, 1080p, RGB+ED with distortion loss, fwd+bwd. The oracle cannot finish
    this size in seconds: check size-independent properties (sortedness, alpha range, normals bounded by alpha, linearity
    of the colour channels, gradient identity sum_g v_feat = sum_px alpha)."""
    import bench

    sc, W, H = bench.make_workload(1_000_000, DEV)
    feats = torch.rand(1_000_000, 3, device=DEV)
    leaves = {k: sc[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities")}
    f = feats.clone().requires_grad_(True)
    rc, ra, rn, sn, rd, rm, meta = G.rasterization_2dgs(leaves["means"], leaves["quats"], leaves["scales"],
                                                        leaves["opacities"], f, sc["viewmats"], sc["Ks"], W, H,
                                                        render_mode="RGB+ED", distloss=True, packed=True)
    ids, fl = meta["isect_ids"], meta["flatten_ids"]
    assert ids.numel() > 1_000_000 and bool((ids[1:] >= ids[:-1]).all())
    same = ids[1:] == ids[:-1]
    assert bool((fl[1:][same] > fl[:-1][same]).all())
    assert rc.shape == (1, H, W, 4) and sn.shape[-3:] == (H, W, 3)
    assert float(ra.min()) >= 0.0 and float(ra.max()) <= 1.0
    for t in (rc, rn, rd, rm):
        assert torch.isfinite(t).all()
    assert bool((rn.norm(dim=-1) <= ra[..., 0] * (1 + 1e-4) + 1e-5).all()), "|sum w n| <= sum w"
    assert bool((rd >= -1e-3).all()), "the L1 distortion is a sum of non-negative pair terms"
    (rc[..., :3].sum() + rd.sum()).backward()
    for k, v in leaves.items():
        assert torch.isfinite(v.grad).all(), k
    assert torch.isfinite(meta["gradient_2dgs"].grad).all()
    # colour channels are linear in the features; alpha / depth do not depend on them
    with torch.no_grad():
        f2 = torch.rand(1_000_000, 3, device=DEV)
        common = (sc["means"], sc["quats"], sc["scales"], sc["opacities"])
        r1 = G.rasterization_2dgs(*common, feats, sc["viewmats"], sc["Ks"], W, H, render_mode="RGB+ED")
        r2 = G.rasterization_2dgs(*common, f2, sc["viewmats"], sc["Ks"], W, H, render_mode="RGB+ED", packed=True)
        r12 = G.rasterization_2dgs(*common, feats + f2, sc["viewmats"], sc["Ks"], W, H, render_mode="RGB+ED")
    assert torch.equal(r1[1], r2[1]) and torch.equal(r1[1], r12[1]), "alpha: packed == dense, independent of colour"
    assert torch.allclose(r12[0][..., :3], r1[0][..., :3] + r2[0][..., :3], rtol=1e-4, atol=1e-4)
    assert torch.allclose(r12[0][..., 3], r1[0][..., 3], rtol=1e-5, atol=1e-5)
    g1 = torch.rand(1_000_000, 1, device=DEV).requires_grad_(True)
    r, a, *_ = G.rasterization_2dgs(*common, g1, sc["viewmats"], sc["Ks"], W, H, packed=True)
    r.sum().backward()
    assert abs(g1.grad.double().sum() - a.double().sum()) <= 1e-3 * a.double().sum()


@pytest.mark.parametrize("packed", [False, True])
def test_rasterization_2dgs_empty_and_invisible_scenes(packed):
    """No surfels / every surfel behind the camera: background images, zero alpha, empty lists, zero gradients."""
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    import gsplat_amd as G
    from _util import make_scene

    W, H, C = 80, 48, 1
    names = ("means", "quats", "scales", "opacities", "colors")
    for empty in (True, False):
        sc, _, _ = make_scene(N=40, C=C, width=W, height=H, seed=5)
        if empty:
            sc = {k: (v[:0] if k in names else v) for k, v in sc.items()}
        else:
            sc["means"] = sc["means"] * torch.tensor([1.0, 1.0, -1.0])
        leaves = {k: sc[k].cuda().clone().requires_grad_(True) for k in names}
        bg = torch.tensor([[0.3, 0.5, 0.7]]).cuda()
        rc, ra, rn, sn, rd, rm, meta = G.rasterization_2dgs(
            leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"],
            sc["viewmats"].cuda(), sc["Ks"].cuda(), W, H, packed=packed, backgrounds=bg, render_mode="RGB+D", distloss=True)
        assert rc.shape == (C, H, W, 4) and float(ra.abs().max()) == 0.0 and meta["isect_ids"].numel() == 0
        assert torch.equal(rc[..., :3], bg[:, None, None, :].expand(C, H, W, 3))
        assert float(rn.abs().max()) == 0.0 and float(rd.abs().max()) == 0.0
        (rc.sum() + ra.sum() + rn.sum() + rd.sum()).backward()
        for k in names:
            g = leaves[k].grad
            assert g is None or float(g.abs().sum()) == 0.0, k


@pytest.mark.parametrize("expected_depth,depth_source", [(True, 1), (False, 1), (True, 2), (True, 0), (False, 0)])
@pytest.mark.parametrize("C,W,H", [(2, 150, 37), (1, 64, 4), (1, 3, 3)])
def test_surfel_post_matches_tensor_ops(G, expected_depth, depth_source, C, W, H):
    """The fused per-pixel tail of rasterization_2dgs (csrc/surfel_post.hip) against the tensor-op composition it replaces
    (the reference's: gsplat/rendering.py:1519-1552, utils.py depth_to_normal with F.normalize), evaluated in float64 on the
    same inputs: outputs and every gradient. Includes pixels with alpha 0 / depth 0 (zero-length cross products)."""
    from gsplat_amd.rendering import _SurfelPost, _depth_to_points

    g = torch.Generator().manual_seed(W * 31 + H)
    D = 4
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    alphas = (0.3 + 0.7 * torch.rand(C, H, W, 1, generator=g))
    depth = 2.0 + 0.02 * xx[None, :, :, None] + 0.03 * yy[None, :, :, None] + 0.2 * torch.rand(C, H, W, 1, generator=g)
    colors = torch.cat([torch.rand(C, H, W, D - 1, generator=g), depth * (alphas if expected_depth else 1.0)], -1)
    median = depth + 0.1 * torch.rand(C, H, W, 1, generator=g)
    if H > 8:  # an empty region: nothing composited
        alphas[:, :6, :9] = 0.0
        colors[:, :6, :9] = 0.0
        median[:, :6, :9] = 0.0
    normals = torch.randn(C, H, W, 3, generator=g)
    sc, _, _ = make_scene(N=4, C=C, width=W, height=H, seed=5)
    viewmats, Ks = sc["viewmats"], sc["Ks"]
    w = [torch.randn(C, H, W, k, generator=g) for k in (D, 3, 3)]

    def reference(colors, alphas, normals, median, dt):
        vm, K = viewmats.to(DEV, dt), Ks.to(DEV, dt)
        if expected_depth:
            colors = torch.cat([colors[..., :-1], colors[..., -1:] / alphas.clamp_min(1e-10)], -1)
        c2w = torch.linalg.inv(vm)
        surf = None
        if depth_source:
            pts = _depth_to_points(median if depth_source == 2 else colors[..., -1:], c2w, K)
            du = pts[..., 2:, 1:-1, :] - pts[..., :-2, 1:-1, :]
            dv = pts[..., 1:-1, 2:, :] - pts[..., 1:-1, :-2, :]
            surf = torch.nn.functional.pad(torch.nn.functional.normalize(torch.linalg.cross(du, dv, dim=-1), dim=-1),
                                           (0, 0, 1, 1, 1, 1))
        return colors, torch.einsum("...ij,...hwj->...hwi", c2w[..., :3, :3], normals), surf

    def loss(outs):
        return sum((o * x.to(o)).sum() for o, x in zip(outs, w) if o is not None and o.numel())

    leaves = [t.to(DEV).clone().requires_grad_(True) for t in (colors, alphas, normals, median)]
    co, nw, sn = _SurfelPost.apply(*leaves, viewmats.to(DEV), Ks.to(DEV), expected_depth, depth_source)
    got = (co if expected_depth else None, nw, sn if depth_source else None)
    loss(got).backward()
    ref_leaves = [t.to(DEV, torch.float64).clone().requires_grad_(True) for t in (colors, alphas, normals, median)]
    want = reference(*ref_leaves, torch.float64)
    loss((want[0] if expected_depth else None, want[1], want[2])).backward()
    for nm, a, b in zip(("colors", "normals_world", "surf_normals"), got, want):
        if a is None:
            continue
        # the normal of a nearly flat patch is a difference of nearly equal products: fp32 carries ~1e-4 of it
        tol = 2e-3 if nm == "surf_normals" else 1e-5
        assert_close_ratio(cpu(a), cpu(b).float(), tol, tol, max_bad_ratio=2e-3 if nm == "surf_normals" else 0.0, name=nm)
    for nm, a, b in zip(("v_colors", "v_alphas", "v_normals", "v_median"), leaves, ref_leaves):
        if b.grad is None:
            assert a.grad is None or float(a.grad.abs().max()) == 0.0, nm
            continue
        assert torch.isfinite(a.grad).all(), nm
        assert_grad_close(cpu(a.grad), cpu(b.grad).float(), rel=5e-3 if depth_source else 1e-5, max_bad_ratio=2e-3, name=nm)


def test_dispatcher_2dgs_bwd_with_dense_cotangents(G):
    """`torch.ops.gsplat.rasterize_to_pixels_2dgs_bwd` through the DISPATCHER (the schema is the reference's: five non-optional
    cotangents) equals the private entry that gsplat_amd's autograd node uses with None for the unused ones."""
    from gsplat_amd import _ops

    N, C, W, H, ts, D = 2000, 1, 128, 96, 16, 4
    sc, W, H = make_scene(N=N, C=C, width=W, height=H, seed=81)
    dsc = {k: v.to(DEV) for k, v in sc.items()}
    rad, m2, d, M, nrm = G.fully_fused_projection_2dgs(dsc["means"], dsc["quats"], dsc["scales"], dsc["viewmats"], dsc["Ks"], W, H)
    op = dsc["opacities"][None].expand(C, -1).contiguous()
    tw, th = math.ceil(W / ts), math.ceil(H / ts)
    _, ids, fl = G.isect_tiles(m2, rad, d, ts, tw, th)
    off = G.isect_offset_encode(ids, C, tw, th)
    g = torch.Generator().manual_seed(3)
    colors = torch.rand(C, N, D, generator=g).to(DEV)
    densify = torch.zeros_like(m2)
    fwd = torch.ops.gsplat.rasterize_to_pixels_2dgs(m2, M, colors, op, nrm, densify, None, None, W, H, ts, off, fl, False, False, True)
    rc, ra, rn, rd, rm, _absgrad, last_ids, median_ids = fwd
    v_rc = torch.randn(rc.shape, generator=g).to(DEV)
    zeros = [torch.zeros_like(t) for t in (ra, rn, rd, rm)]
    head = (m2, M, colors, op, nrm, densify, None, None, off, fl, rc, ra, last_ids, median_ids, W, H, ts, False, v_rc)
    dense = torch.ops.gsplat.rasterize_to_pixels_2dgs_bwd(*head, *zeros, False)
    private = _ops.impl("rasterize_to_pixels_2dgs_bwd")(*head, None, None, None, None, False)
    for a, b in zip(dense, private):
        assert (a is None) == (b is None)
        if a is not None:  # two launches add the per-tile sums to the gradient rows in different orders: equal to rounding
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-9
