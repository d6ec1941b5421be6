"""GPU parity: every stage op (through torch.ops.gsplat.* -> C-ABI -> HIP kernels) against the CPU oracle
on the same seeded inputs, and against the committed golden vectors produced by the reference's Python.

Tolerances are the reference's own (tests/test_basic.py:427-504, 1271-1315, 2639-2692):
integer outputs exact; fp32 forward rtol 1e-4/atol 1e-4 (projection), default fp32 closeness for renders;
gradients scale-relative (atomics accumulate in unspecified fp32 order)."""
import math

import numpy as np
import pytest
import torch

from _util import assert_close_ratio, assert_grad_close, make_scene, to_t

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


# ------------------------------------------------------------------------------------------------
def test_library_is_native_gfx950(G):
    from gsplat_amd import _cabi

    assert _cabi.ARCH == "gfx950"
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


@pytest.mark.parametrize("triu", [False, True])
def test_quat_scale_to_covar_preci(G, O, triu):
    N = 1000
    q = torch.randn(N, 4)
    s = torch.rand(N, 3) * 0.5 + 0.05
    qg, sg = q.to(DEV).requires_grad_(True), s.to(DEV).requires_grad_(True)
    c, p = G.quat_scale_to_covar_preci(qg, sg, True, True, triu)
    qo, so = q.clone().requires_grad_(True), s.clone().requires_grad_(True)
    c_o, p_o = O.quat_scale_to_covar_preci(qo, so, True, True, triu)
    assert_close_ratio(cpu(c), c_o, 1e-5, 1e-6, name="covars")
    assert_close_ratio(cpu(p), p_o, 1e-4, 1e-3, name="precis")
    wc, wp = torch.randn_like(c_o), torch.randn_like(p_o) * 1e-3
    (c * wc.to(DEV)).sum().add((p * wp.to(DEV)).sum()).backward()
    (c_o * wc).sum().add((p_o * wp).sum()).backward()
    assert_grad_close(cpu(qg.grad), qo.grad, rel=1e-4, name="v_quats")
    assert_grad_close(cpu(sg.grad), so.grad, rel=1e-4, name="v_scales")
    only_c, none_p = G.quat_scale_to_covar_preci(qg, sg, True, False, triu)
    assert none_p is None and only_c.shape == c.shape


@pytest.mark.parametrize("triu", [False, True])
def test_quat_scale_to_covar_preci_float64(G, O, triu):
    """The reference instantiates this op for double as well (QuatScaleToCovarCUDA.cu:145) and its tests pass float64: the
    double-precision kernels (csrc/quat_scale_f64.hip) against the oracle's torch formulas evaluated in float64, forward and
    gradients, to double rounding; mixed dtypes are refused."""
    N = 777
    g = torch.Generator().manual_seed(5)
    q = torch.randn(N, 4, generator=g, dtype=torch.float64)
    s = torch.rand(N, 3, generator=g, dtype=torch.float64) * 0.5 + 0.05
    qg, sg = q.to(DEV).requires_grad_(True), s.to(DEV).requires_grad_(True)
    c, p = G.quat_scale_to_covar_preci(qg, sg, True, True, triu)
    assert c.dtype == torch.float64 and p.dtype == torch.float64
    qo, so = q.clone().requires_grad_(True), s.clone().requires_grad_(True)
    c_o, p_o = O.quat_scale_to_covar_preci(qo, so, True, True, triu)
    assert c_o.dtype == torch.float64
    torch.testing.assert_close(cpu(c), c_o, rtol=1e-12, atol=1e-13)
    torch.testing.assert_close(cpu(p), p_o, rtol=1e-11, atol=1e-11)
    wc, wp = torch.randn(c_o.shape, generator=g, dtype=torch.float64), torch.randn(p_o.shape, generator=g, dtype=torch.float64) * 1e-3
    (c * wc.to(DEV)).sum().add((p * wp.to(DEV)).sum()).backward()
    (c_o * wc).sum().add((p_o * wp).sum()).backward()
    torch.testing.assert_close(cpu(qg.grad), qo.grad, rtol=1e-9, atol=1e-10)
    torch.testing.assert_close(cpu(sg.grad), so.grad, rtol=1e-9, atol=1e-10)
    with pytest.raises(TypeError):
        G.quat_scale_to_covar_preci(qg.detach(), sg.detach().float(), True, True, triu)


def _proj_both(G, O, sc, W, H, cam, use_covars=False, opac=True, comp=True, radius_clip=0.0, grads=True):
    names = ("means", "quats", "scales", "viewmats")
    tg = {k: sc[k].to(DEV).clone().requires_grad_(grads) for k in names}
    to = {k: sc[k].clone().requires_grad_(grads) for k in names}
    ops_g = sc["opacities"].to(DEV) if opac else None
    ops_o = sc["opacities"][None] if opac else None
    cov_g = cov_o = None
    if use_covars:
        cov_o6 = O.quat_scale_to_covar_preci(to["quats"], to["scales"], True, False, True)[0]
        cov_g6 = G.quat_scale_to_covar_preci(tg["quats"], tg["scales"], True, False, True)[0]
        cov_g, cov_o = cov_g6, cov_o6[None]
    out_g = G.fully_fused_projection(tg["means"], cov_g, None if use_covars else tg["quats"],
                                     None if use_covars else tg["scales"], tg["viewmats"], sc["Ks"].to(DEV), W, H,
                                     eps2d=0.3, near_plane=0.01, far_plane=1e10, radius_clip=radius_clip, packed=False,
                                     calc_compensations=comp, camera_model=cam, opacities=ops_g)
    out_o = O.fully_fused_projection(to["means"][None], cov_o, None if use_covars else to["quats"][None],
                                     None if use_covars else to["scales"][None], to["viewmats"][None], sc["Ks"][None],
                                     W, H, 0.3, 0.01, 1e10, radius_clip, comp, cam, ops_o)
    out_o = [None if o is None else o[0] for o in out_o]
    return tg, to, out_g, out_o


@pytest.mark.parametrize("cam", ["pinhole", "ortho", "fisheye"])
@pytest.mark.parametrize("use_covars", [False, True])
def test_projection_dense_fwd_bwd(G, O, cam, use_covars):
    sc, W, H = make_scene(N=4000, C=3, width=200, height=150, seed=1)
    if cam == "ortho":
        sc["Ks"][:, 0, 0] = sc["Ks"][:, 1, 1] = 20.0
    tg, to, (rad, m2, d, con, comp), (rad_o, m2_o, d_o, con_o, comp_o) = _proj_both(G, O, sc, W, H, cam, use_covars)
    vis_g, vis_o = (cpu(rad) > 0).all(-1), (rad_o > 0).all(-1)
    assert (vis_g == vis_o).float().mean() > 0.999, "visibility decisions differ"
    valid = vis_g & vis_o
    assert valid.sum() > 500
    assert (cpu(rad)[valid] - rad_o[valid]).abs().max() <= 1
    assert_close_ratio(cpu(m2)[valid], m2_o[valid], 1e-4, 1e-4, name="means2d")
    assert_close_ratio(cpu(d)[valid], d_o[valid], 1e-4, 1e-4, name="depths")
    assert_close_ratio(cpu(con)[valid], con_o[valid], 1e-4, 1e-4, name="conics")
    assert_close_ratio(cpu(comp)[valid], comp_o[valid], 1e-4, 1e-3, name="compensations")
    # culled rows are zero-filled (deterministic, unlike the reference's uninitialised memory)
    assert cpu(m2)[~vis_g].abs().max() == 0 and cpu(con)[~vis_g].abs().max() == 0
    # gradients: same random cotangents, restricted to commonly-valid rows
    vm = valid[..., None].float()
    w2, wd, wc, wk = (torch.randn_like(m2_o), torch.randn_like(d_o), torch.randn_like(con_o) * 1e-2,
                      torch.randn_like(comp_o))
    loss_o = (m2_o * w2 * vm).sum() + (d_o * wd * valid).sum() + (con_o * wc * vm).sum() + (comp_o * wk * valid).sum()
    loss_g = ((m2 * (w2 * vm).to(DEV)).sum() + (d * (wd * valid).to(DEV)).sum() + (con * (wc * vm).to(DEV)).sum()
              + (comp * (wk * valid).to(DEV)).sum())
    loss_o.backward()
    loss_g.backward()
    for k in ("means", "quats", "scales", "viewmats"):
        assert_grad_close(cpu(tg[k].grad), to[k].grad, rel=2e-3, name=f"v_{k}")


@pytest.mark.parametrize("packed", [False, True])
def test_projection_double_instantiation(G, packed):
    """The reference dispatches the projection ops over float AND double (ProjectionEWA3DGSFused.cu:260, 686,
    ProjectionEWA3DGSPacked.cu:344, 733); its double instantiation keeps doubles in MEMORY only - every value is loaded into
    glm float vectors (include/Common.h:65-70), the arithmetic is float, results are widened on store. So: float64 inputs give
    float64 outputs and gradients that EQUAL the float32 run of the float-rounded inputs, bit for bit."""
    sc, W, H = make_scene(N=3000, C=2, width=160, height=120, seed=12)
    names = ("means", "quats", "scales", "viewmats")
    d64 = {k: sc[k].double().to(DEV).requires_grad_(True) for k in names}
    f32 = {k: sc[k].to(DEV).clone().requires_grad_(True) for k in names}
    kw = dict(eps2d=0.3, near_plane=0.01, far_plane=1e10, radius_clip=0.0, packed=packed, calc_compensations=True,
              camera_model="pinhole")
    out64 = G.fully_fused_projection(d64["means"], None, d64["quats"], d64["scales"], d64["viewmats"], sc["Ks"].double().to(DEV),
                                     W, H, opacities=sc["opacities"].double().to(DEV), **kw)
    out32 = G.fully_fused_projection(f32["means"], None, f32["quats"], f32["scales"], f32["viewmats"], sc["Ks"].to(DEV), W, H,
                                     opacities=sc["opacities"].to(DEV), **kw)
    assert len(out64) == len(out32)
    g = torch.Generator().manual_seed(4)
    l64 = l32 = 0.0
    for a, b in zip(out64, out32):
        if a.is_floating_point():
            assert a.dtype == torch.float64 and b.dtype == torch.float32
            assert torch.equal(a, b.double())
            w = torch.randn(b.shape, generator=g).to(DEV)
            l64, l32 = l64 + (a * w.double()).sum(), l32 + (b * w).sum()
        else:
            assert a.dtype == b.dtype and torch.equal(a, b)
    l64.backward()
    l32.backward()
    for k in names:
        assert d64[k].grad.dtype == torch.float64
        if k == "viewmats":  # summed over the Gaussians with float atomics: the order of two launches may differ
            torch.testing.assert_close(d64[k].grad, f32[k].grad.double(), rtol=1e-5, atol=1e-3)
        else:
            assert torch.equal(d64[k].grad, f32[k].grad.double()), k


def test_projection_culling_rules(G, O):
    sc, W, H = make_scene(N=3000, C=2, width=160, height=120, seed=2, z_range=(0.5, 30.0))
    sc["opacities"][:500] = 0.002  # below 1/255 -> culled when opacities are passed
    _, _, (rad, *_), (rad_o, *_) = _proj_both(G, O, sc, W, H, "pinhole", opac=True, comp=False, radius_clip=2.0,
                                              grads=False)
    vis_g, vis_o = (cpu(rad) > 0).all(-1), (rad_o > 0).all(-1)
    assert (vis_g == vis_o).float().mean() > 0.999
    assert not vis_g[:, :500].any()
    big = (cpu(rad) > 0).all(-1)
    assert (cpu(rad)[big].max(-1).values > 2).all(), "radius_clip must drop splats with both radii <= clip"


@pytest.mark.parametrize("sparse_grad,C", [(False, 3), (True, 3), (True, 1)])
def test_projection_packed_matches_dense(G, sparse_grad, C):
    sc, W, H = make_scene(N=5000, C=C, width=200, height=150, seed=3)
    a = {k: v.to(DEV) for k, v in sc.items()}
    leaves_d = [a[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "viewmats")]
    leaves_p = [a[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "viewmats")]
    kw = dict(eps2d=0.3, near_plane=0.01, far_plane=1e10, radius_clip=0.0, calc_compensations=True,
              camera_model="pinhole", opacities=a["opacities"])
    rad, m2, d, con, comp = G.fully_fused_projection(leaves_d[0], None, leaves_d[1], leaves_d[2], leaves_d[3], a["Ks"],
                                                     W, H, packed=False, **kw)
    bi, ci, gi, indptr, rad_p, m2_p, d_p, con_p, comp_p = G.fully_fused_projection(
        leaves_p[0], None, leaves_p[1], leaves_p[2], leaves_p[3], a["Ks"], W, H, packed=True, sparse_grad=sparse_grad,
        **kw)
    vis = (rad > 0).all(-1)
    ci_d, gi_d = torch.where(vis)
    assert bi.dtype == torch.int64 and (bi == 0).all()
    assert torch.equal(ci, ci_d) and torch.equal(gi, gi_d), "packed rows must be ordered by (camera, gaussian)"
    assert torch.equal(rad_p, rad[vis]) and torch.equal(m2_p, m2[vis]) and torch.equal(con_p, con[vis])
    assert torch.equal(d_p, d[vis]) and torch.equal(comp_p, comp[vis])
    counts = vis.sum(-1).cumsum(0)
    assert indptr.dtype == torch.int32 and indptr[0] == 0 and torch.equal(indptr[1:].long(), counts)
    w2, wd, wc, wk = torch.randn_like(m2_p), torch.randn_like(d_p), torch.randn_like(con_p) * 1e-2, torch.randn_like(comp_p)
    ((m2_p * w2).sum() + (d_p * wd).sum() + (con_p * wc).sum() + (comp_p * wk).sum()).backward()
    ((m2[vis] * w2).sum() + (d[vis] * wd).sum() + (con[vis] * wc).sum() + (comp[vis] * wk).sum()).backward()
    for nm, lp, ld in zip(("means", "quats", "scales", "viewmats"), leaves_p, leaves_d):
        g = lp.grad
        if sparse_grad and nm != "viewmats":
            # the reference's layout (Projection.cpp:1125-1200): COO over the Gaussian axis, one entry per packed row in
            # packed-row order, coalesced iff a single image
            assert g.is_sparse and g.shape == ld.shape and g._nnz() == gi.numel(), nm
            assert torch.equal(g._indices(), gi[None]), nm
            g = g.to_dense()
        else:
            assert not g.is_sparse, nm
        assert_grad_close(cpu(g), cpu(ld.grad), rel=1e-4, name=f"packed v_{nm}")
    if sparse_grad:
        # the op itself (autograd's accumulation does not keep the flag): coalesced iff a single image, like the reference
        out = torch.ops.gsplat.projection_ewa_3dgs_packed_bwd(
            a["means"], None, a["quats"], a["scales"], a["viewmats"], a["Ks"], W, H, 0.3, 0, True, bi, ci, gi, con_p.detach(),
            comp_p.detach(), w2, wd, wc, wk, False)
        assert out[0].is_sparse and out[0].is_coalesced() == (C == 1) and out[1] is None
        assert out[2].is_sparse and out[3].is_sparse and out[4] is None


def test_projection_packed_sparse_grad_allocates_no_dense_rows(G):
    """sparse_grad exists to keep the backward's memory proportional to the visible rows (the reference's 49 M / 107 M
    Gaussian profiles are packed + sparse_grad, docs/source/tests/profile.rst:127,143): with 1 % of 2 M Gaussians in view the
    backward must stay far below one dense [N, 10] gradient set (80 MB)."""
    N = 2_000_000
    g = torch.Generator().manual_seed(5)
    means = torch.randn(N, 3, generator=g) * 0.5
    means[:, 2] += 5.0
    means[N // 100:, 2] = -5.0  # behind the camera: culled
    quats = torch.nn.functional.normalize(torch.randn(N, 4, generator=g), dim=-1)
    scales = torch.rand(N, 3, generator=g) * 0.02 + 0.005
    viewmats = torch.eye(4)[None]
    Ks = torch.tensor([[[300.0, 0, 128], [0, 300.0, 128], [0, 0, 1]]])
    lv = [t.to(DEV).requires_grad_(True) for t in (means, quats, scales)]
    out = G.fully_fused_projection(lv[0], None, lv[1], lv[2], viewmats.to(DEV), Ks.to(DEV), 256, 256, packed=True,
                                   sparse_grad=True)
    m2_p, d_p, con_p = out[5], out[6], out[7]
    nnz = m2_p.shape[0]
    assert 0 < nnz <= N // 100
    loss = m2_p.sum() + d_p.sum() + con_p.sum()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    loss.backward()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    assert peak < 16 * 2 ** 20, f"sparse_grad backward peaked at {peak / 2**20:.1f} MiB above the forward state"
    for t in lv:
        assert t.grad.is_sparse and t.grad._nnz() == nnz


# K * 3 floats per row a multiple of four: rows move as wave-cooperative tiles (K == bands in use, or with unused bands
# behind them); K = 25: 75 floats per row, every thread streams its own row
@pytest.mark.parametrize("deg,K", [(0, 25), (1, 25), (2, 25), (3, 25), (4, 25), (3, 16), (1, 16), (2, 28), (0, 4), (1, 4),
                                   (4, 28)])
def test_spherical_harmonics_dense(G, O, deg, K):
    sc, W, H = make_scene(N=3000, C=3, seed=4)
    coeffs = torch.randn(3000, K, 3) * 0.3
    masks = torch.rand(3, 3000) > 0.2
    mg = sc["means"].to(DEV).requires_grad_(True)
    cg = coeffs.to(DEV).requires_grad_(True)
    col = G.spherical_harmonics(deg, mg, sc["viewmats"].to(DEV), cg, masks=masks.to(DEV))
    mo, co = sc["means"].clone().requires_grad_(True), coeffs.clone().requires_grad_(True)
    col_o = O.spherical_harmonics(deg, mo[None], sc["viewmats"][None], co, masks[None])[0]
    assert_close_ratio(cpu(col), col_o, 1e-5, 1e-5, name="sh colors")
    w = torch.randn_like(col_o)
    (col * w.to(DEV)).sum().backward()
    (col_o * w).sum().backward()
    assert_grad_close(cpu(cg.grad), co.grad, rel=1e-5, name="v_coeffs")
    if deg > 0:
        assert_grad_close(cpu(mg.grad), mo.grad, rel=1e-4, name="v_means")
    else:
        assert cpu(mg.grad).abs().max() < 1e-6


def test_spherical_harmonics_packed(G, O):
    sc, W, H = make_scene(N=2000, C=2, seed=5)
    coeffs = torch.randn(2000, 16, 3) * 0.3
    vis = torch.rand(2, 2000) > 0.5
    ci, gi = torch.where(vis)
    bi = torch.zeros_like(ci)
    mg = sc["means"].to(DEV).requires_grad_(True)
    cg = coeffs.to(DEV).requires_grad_(True)
    col = G.spherical_harmonics(3, mg, sc["viewmats"].to(DEV), cg[gi.to(DEV)], batch_ids=bi.to(DEV),
                                camera_ids=ci.to(DEV), gaussian_ids=gi.to(DEV))
    mo, co = sc["means"].clone().requires_grad_(True), coeffs.clone().requires_grad_(True)
    col_o = O.spherical_harmonics(3, mo[None], sc["viewmats"][None], co)[0][vis]
    assert_close_ratio(cpu(col), col_o, 1e-5, 1e-5, name="packed sh colors")
    w = torch.randn_like(col_o)
    (col * w.to(DEV)).sum().backward()
    (col_o * w).sum().backward()
    assert_grad_close(cpu(cg.grad), co.grad, rel=1e-5, name="packed v_coeffs")
    assert_grad_close(cpu(mg.grad), mo.grad, rel=1e-4, name="packed v_means")


@pytest.mark.parametrize("D", [3, 5])  # D = 3: one-thread-per-row kernels; other widths: per-(row, channel) kernels
@pytest.mark.parametrize("packed", [False, True])
def test_spherical_harmonics_viewmat_gradient(G, O, D, packed):
    """Pose gradient of the SH colours (reference SphericalHarmonicsViewDirectionCUDA.cu; autograd of the torch
    restatement _torch_impl.py:1052-1067 is the oracle): v_viewmats from the per-row d(loss)/d(direction)."""
    sc, W, H = make_scene(N=1500, C=3, seed=21)
    coeffs = torch.randn(1500, 16, D) * 0.3
    vm_g = sc["viewmats"].to(DEV).requires_grad_(True)
    mg = sc["means"].to(DEV).requires_grad_(True)
    cg = coeffs.to(DEV).requires_grad_(True)
    vm_o = sc["viewmats"].clone().requires_grad_(True)
    mo, co = sc["means"].clone().requires_grad_(True), coeffs.clone().requires_grad_(True)
    if packed:
        vis = torch.rand(3, 1500) > 0.4
        ci, gi = torch.where(vis)
        col = G.spherical_harmonics(3, mg, vm_g, cg[gi.to(DEV)], batch_ids=torch.zeros_like(ci).to(DEV),
                                    camera_ids=ci.to(DEV), gaussian_ids=gi.to(DEV))
        col_o = O.spherical_harmonics(3, mo[None], vm_o[None], co)[0][vis]
    else:
        masks = torch.rand(3, 1500) > 0.2
        col = G.spherical_harmonics(3, mg, vm_g, cg, masks=masks.to(DEV))
        col_o = O.spherical_harmonics(3, mo[None], vm_o[None], co, masks[None])[0]
    assert_close_ratio(cpu(col), col_o, 1e-5, 1e-5, name="sh colors")
    w = torch.randn_like(col_o)
    (col * w.to(DEV)).sum().backward()
    (col_o * w).sum().backward()
    assert_grad_close(cpu(vm_g.grad)[:, :3], vm_o.grad[:, :3], rel=2e-4, name="v_viewmats")
    assert cpu(vm_g.grad)[:, 3].abs().max() == 0  # the last row of a view matrix carries no gradient
    assert_grad_close(cpu(mg.grad), mo.grad, rel=1e-4, name="v_means")
    assert_grad_close(cpu(cg.grad), co.grad, rel=1e-5, name="v_coeffs")


# ------------------------------------------------------------------------------------------------
def _project_scene(G, sc, W, H):
    a = {k: v.to(DEV) for k, v in sc.items()}
    rad, m2, d, con, _ = G.fully_fused_projection(a["means"], None, a["quats"], a["scales"], a["viewmats"], a["Ks"],
                                                  W, H, opacities=a["opacities"])
    op = a["opacities"][None].expand(m2.shape[0], -1).contiguous()
    return a, rad, m2, d, con, op


@pytest.mark.parametrize("mode", ["aabb", "ellipse"])
@pytest.mark.parametrize("tile_size", [16, 8])
def test_isect_exact_dense(G, O, mode, tile_size):
    sc, W, H = make_scene(N=20000, C=3, width=320, height=200, seed=6)
    a, rad, m2, d, con, op = _project_scene(G, sc, W, H)
    tw, th = math.ceil(W / tile_size), math.ceil(H / tile_size)
    kw = dict(conics=con, opacities=op) if mode == "ellipse" else {}
    tpg, ids, fl = G.isect_tiles(m2, rad, d, tile_size, tw, th, **kw)
    kwo = dict(conics=cpu(con), opacities=cpu(op)) if mode == "ellipse" else {}
    tpg_o, ids_o, fl_o = O.isect_tiles(cpu(m2), cpu(rad), cpu(d), tile_size, tw, th, **kwo)
    assert tpg.dtype == torch.int32 and ids.dtype == torch.int64 and fl.dtype == torch.int32
    assert torch.equal(cpu(tpg), tpg_o), "tiles_per_gauss must be bit-exact"
    assert torch.equal(cpu(ids), ids_o), "sorted isect_ids must be bit-exact"
    assert torch.equal(cpu(fl), fl_o), "flatten_ids must be bit-exact (stable sort)"
    off = G.isect_offset_encode(ids, 3, tw, th)
    assert torch.equal(cpu(off), O.isect_offset_encode(ids_o, 3, tw, th))
    # unsorted emission order is part of the contract too
    _, ids_u, fl_u = G.isect_tiles(m2, rad, d, tile_size, tw, th, sort=False, **kw)
    _, ids_uo, fl_uo = O.isect_tiles(cpu(m2), cpu(rad), cpu(d), tile_size, tw, th, sort=False, **kwo)
    assert torch.equal(cpu(ids_u), ids_uo) and torch.equal(cpu(fl_u), fl_uo)


_ISECT_CASES = {
    # name: (N, C, width, height, tile_size, mode, variant)
    "ellipse-3img": (20000, 3, 320, 200, 16, "ellipse", None),
    "aabb-3img-ts8": (20000, 3, 320, 200, 8, "aabb", None),
    "ellipse-ts4": (3000, 2, 160, 112, 4, "ellipse", None),
    "packed": (20000, 1, 320, 200, 16, "ellipse", "packed"),
    "depth-ties": (20000, 2, 320, 200, 16, "ellipse", "ties"),
    "one-depth": (6000, 1, 320, 200, 16, "aabb", "flat"),
    # depths that do not order like positive normal doubles: those lists leave the f64 sorting network (bitonic64.hpp)
    "odd-depths": (20000, 2, 320, 200, 16, "ellipse", "odd"),
    "odd-depths-long-tiles": (40000, 1, 640, 360, 16, "aabb", "odd-cluster"),
    "cluster-long-tiles": (40000, 1, 640, 360, 16, "ellipse", "cluster"),  # tiles beyond the LDS arena: work-list sort
    "giants-retry": (20000, 1, 640, 360, 16, "ellipse", "giant"),  # more (row, bin) entries than the workspace: retry path
}


def _odd_depths(d, seed=5):
    """A few per cent of the depths replaced by values whose float bits are no positive normal double's high word: negative,
    zero, denormal, +inf, NaN (one quiet pattern: the key carries the bits). Ties among them included."""
    g = torch.Generator().manual_seed(seed)
    d = d.clone()
    flat = d.view(-1)
    pick = torch.rand(flat.numel(), generator=g).to(d.device)
    kind = torch.randint(0, 5, (flat.numel(),), generator=g).to(d.device)
    odd = pick < 0.04
    vals = torch.tensor([-1.5, 0.0, 1e-42, float("inf"), float("nan")], device=d.device)
    flat[odd] = vals[kind[odd]]
    flat[odd & (kind == 0)] *= torch.randint(1, 4, (int((odd & (kind == 0)).sum()),), generator=g).to(d.device).float()
    return d


@pytest.mark.parametrize("sort", ["f64", "int"])
@pytest.mark.parametrize("path", ["binned", "legacy"])
@pytest.mark.parametrize("case", ["odd-depths", "depth-ties", "cluster-long-tiles", "aabb-3img-ts8"])
def test_isect_sort_networks_match_oracle(G, O, monkeypatch, path, case, sort):
    """GSX_ISECT_SORT=int keeps every tile list on the integer compare-exchange network; the default sorts the same words as
    doubles. Both are the oracle's order, bit for bit."""
    if sort == "int":
        monkeypatch.setenv("GSX_ISECT_SORT", "int")
    test_isect_paths_match_oracle(G, O, monkeypatch, path, case)


@pytest.mark.parametrize("shape", ["2x2", "8x2", "4x4", "3x5", "16x1"])
@pytest.mark.parametrize("case", ["giants-retry", "ellipse-ts4", "aabb-3img-ts8", "cluster-long-tiles", "depth-ties"])
def test_isect_binned_bin_shapes_match_oracle(G, O, monkeypatch, case, shape):
    """GSX_ISECT_BIN=WxH: bins other than the default 4 x 2 tiles run the row kernels' generic instantiation (block records of
    2 x 4 or 2 x 2 bins, csrc/isect_binned.hip: BinShape<false>) - the oracle's lists, bit for bit, whatever the shape."""
    monkeypatch.setenv("GSX_ISECT_BIN", shape)
    test_isect_paths_match_oracle(G, O, monkeypatch, "binned", case)


def test_isect_remembers_an_input_that_sent_the_binned_path_back(G, O):
    """A clustered scene fails the binned path's skew test and the call starts over Gaussian-major; the next calls of that shape
    must not pay for the attempt again (gsx_isect_binned_note_retry), and every call returns the oracle's lists."""
    from gsplat_amd import _cabi
    sc, W, H = make_scene(N=300000, C=1, width=1280, height=720, seed=4)
    sc["means"][:, :2] *= 0.12
    a, rad, m2, d, con, op = _project_scene(G, sc, W, H)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    if not _cabi._lib.gsx_isect_binned_supported(rad.numel() // 2, 1, tw, th, 0):
        pytest.skip("this shape was sent back earlier in this process")
    tried = []
    for _ in range(3):
        _cabi.profile_begin()
        tpg, ids, fl = G.isect_tiles(m2, rad, d, 16, tw, th, conics=con, opacities=op)
        torch.cuda.synchronize()
        tried.append(any("binned" in k for k in _cabi.profile_end()))
    assert tried[0], "the first call tries the tile-owner-major path"
    tpg_o, ids_o, fl_o = O.isect_tiles(cpu(m2), cpu(rad), cpu(d), 16, tw, th, conics=cpu(con), opacities=cpu(op))
    assert torch.equal(cpu(ids), ids_o) and torch.equal(cpu(fl), fl_o) and torch.equal(cpu(tpg), tpg_o)
    # the scene is crowded enough to be sent back (x0.12: bins of > 10 k entries): no second attempt
    assert tried[1:] == [False, False], tried


@pytest.mark.parametrize("case", ["giants-retry", "ellipse-ts4", "aabb-3img-ts8", "cluster-long-tiles", "packed", "depth-ties"])
def test_isect_fused_span_records_match_oracle(G, O, monkeypatch, case):
    """Very large inputs (c4: 16 M rows) keep, per row, the spans the counting pass's walk found, and the emission reads them
    back instead of walking again (csrc/isect_fused.hip). GSX_FUSED_SPANS=1 switches that on at test sizes: giants overflow the
    16-byte record and walk again, tile sizes 4 / 8 change the spans' lengths, ... - the oracle's lists, bit for bit."""
    monkeypatch.setenv("GSX_FUSED_SPANS", "1")
    test_isect_paths_match_oracle(G, O, monkeypatch, "legacy", case)


@pytest.mark.parametrize("path", ["binned", "legacy"])
@pytest.mark.parametrize("case", sorted(_ISECT_CASES))
def test_isect_paths_match_oracle(G, O, monkeypatch, path, case):
    """Both implementations of isect_tiles(sort=True) - tile-owner-major (csrc/isect_binned.hip) and Gaussian-major
    (csrc/isect_fused.hip), normally chosen by density - are forced in turn (GSX_ISECT_PATH) and must equal the C oracle
    bit for bit: tiles_per_gauss, sorted keys, row ids, offsets."""
    N, C, W, H, ts, mode, variant = _ISECT_CASES[case]
    sc, W, H = make_scene(N=N, C=C, width=W, height=H, seed=11)
    if variant in ("cluster", "odd-cluster"):
        sc["means"][:, :2] *= 0.05
    if variant == "giant":
        sc["scales"] = torch.full_like(sc["scales"], 0.4)
    a, rad, m2, d, con, op = _project_scene(G, sc, W, H)
    if variant == "ties":
        d = (d * 2).round() / 2 + 0.25
    if variant == "flat":
        d = torch.ones_like(d)
    if variant in ("odd", "odd-cluster"):
        d = _odd_depths(d)
    tw, th = math.ceil(W / ts), math.ceil(H / ts)
    kw, kwo = {}, {}
    if variant == "packed":
        vis = (rad[0] > 0).all(-1)
        gi = torch.where(vis)[0]
        m2, rad, d, con, op = m2[0][vis], rad[0][vis], d[0][vis], con[0][vis], op[0][vis]
        kw = dict(packed=True, n_images=1, image_ids=torch.zeros_like(gi), gaussian_ids=gi)
        kwo = dict(image_ids=torch.zeros_like(gi).cpu(), n_images=1)
    if mode == "ellipse":
        kw.update(conics=con, opacities=op)
        kwo.update(conics=cpu(con), opacities=cpu(op))
    monkeypatch.setenv("GSX_ISECT_PATH", path)
    tpg, ids, fl = G.isect_tiles(m2, rad, d, ts, tw, th, **kw)
    off = G.isect_offset_encode(ids, C, tw, th)
    monkeypatch.delenv("GSX_ISECT_PATH")
    tpg_o, ids_o, fl_o = O.isect_tiles(cpu(m2), cpu(rad), cpu(d), ts, tw, th, **kwo)
    assert torch.equal(cpu(tpg), tpg_o), "tiles_per_gauss must be bit-exact"
    assert torch.equal(cpu(ids), ids_o), "sorted isect_ids must be bit-exact"
    assert torch.equal(cpu(fl), fl_o), "flatten_ids must be bit-exact (stable sort)"
    assert torch.equal(cpu(off), O.isect_offset_encode(ids_o, C, tw, th))


def test_isect_packed_and_edge_cases(G, O):
    sc, W, H = make_scene(N=6000, C=2, width=200, height=120, seed=7)
    a, rad, m2, d, con, op = _project_scene(G, sc, W, H)
    vis = (rad > 0).all(-1)
    ci, gi = torch.where(vis)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    tpg, ids, fl = G.isect_tiles(m2[vis], rad[vis], d[vis], 16, tw, th, packed=True, n_images=2, image_ids=ci,
                                 gaussian_ids=gi, conics=con[vis], opacities=op[vis])
    tpg_o, ids_o, fl_o = O.isect_tiles(cpu(m2[vis]), cpu(rad[vis]), cpu(d[vis]), 16, tw, th, conics=cpu(con[vis]),
                                       opacities=cpu(op[vis]), image_ids=cpu(ci), n_images=2)
    assert torch.equal(cpu(tpg), tpg_o) and torch.equal(cpu(ids), ids_o) and torch.equal(cpu(fl), fl_o)
    # packed and dense describe the same (image, tile, depth) multiset
    _, ids_d, _ = G.isect_tiles(m2, rad, d, 16, tw, th, conics=con, opacities=op)
    assert torch.equal(ids, ids_d)
    # nothing visible -> empty lists, zero offsets
    z = torch.zeros_like(rad)
    tpg0, ids0, fl0 = G.isect_tiles(m2, z, d, 16, tw, th)
    assert tpg0.sum() == 0 and ids0.numel() == 0 and fl0.numel() == 0
    off0 = G.isect_offset_encode(ids0, 2, tw, th)
    assert off0.shape == (2, th, tw) and off0.abs().sum() == 0
    # zero Gaussians
    e = torch.empty(1, 0, 2, device=DEV)
    tpg_e, ids_e, _ = G.isect_tiles(e, e.int(), torch.empty(1, 0, device=DEV), 16, tw, th)
    assert tpg_e.shape == (1, 0) and ids_e.numel() == 0
    # key-width overflow is rejected like the reference (Intersect.cpp:219-228)
    with pytest.raises(RuntimeError):
        G.isect_tiles(m2[:1], rad[:1], d[:1], 1, 70000, 70000)


@pytest.mark.parametrize("n", [1, 63, 4096, 4097, 100_003, 3_000_000])
def test_radix_sort_matches_stable_sort(G, n):
    from gsplat_amd import _cabi

    g = torch.Generator(device="cpu").manual_seed(n)
    # few distinct keys -> many ties -> exercises stability
    keys = torch.randint(0, 1 << 20, (n,), generator=g, dtype=torch.int64) << 25
    keys |= torch.randint(0, 4, (n,), generator=g, dtype=torch.int64)
    vals = torch.arange(n, dtype=torch.int32)
    k, v = keys.to(DEV), vals.to(DEV)
    k2, v2 = torch.empty_like(k), torch.empty_like(v)
    ws = torch.empty(_cabi.sort_workspace_bytes(n), dtype=torch.uint8, device=DEV)
    in_alt = _cabi.sort_pairs(k, v, k2, v2, n, 46, ws)
    ks, vs = (k2, v2) if in_alt else (k, v)
    order = np.argsort(keys.numpy(), kind="stable")
    assert torch.equal(cpu(ks), keys[order]) and torch.equal(cpu(vs), vals[order])


@pytest.mark.parametrize("n", [1, 255, 4096, 4097, 1_000_001])
def test_scan(G, n):
    from gsplat_amd._ops import _scan_i32

    x = torch.randint(0, 50, (n,), dtype=torch.int32)
    assert torch.equal(cpu(_scan_i32(x.to(DEV))), torch.cumsum(x.long(), 0))


# ------------------------------------------------------------------------------------------------
def _raster_case(G, O, N, C, W, H, tile_size, D, seed, bg=False, masks=False, absgrad=False, packed=False):
    sc, W, H = make_scene(N=N, C=C, width=W, height=H, seed=seed)
    a, rad, m2, d, con, op = _project_scene(G, sc, W, H)
    tw, th = math.ceil(W / tile_size), math.ceil(H / tile_size)
    g = torch.Generator().manual_seed(seed)
    colors = torch.rand(C, N, D, generator=g).to(DEV)
    backgrounds = torch.rand(C, D, generator=g).to(DEV) if bg else None
    tile_masks = (torch.rand(C, th, tw, generator=g) > 0.3).to(DEV) if masks else None
    if packed:
        vis = (rad > 0).all(-1)
        ci, gi = torch.where(vis)
        m2, con, op, colors, radp, dp = m2[vis], con[vis], op[vis], colors[vis], rad[vis], d[vis]
        _, ids, fl = G.isect_tiles(m2, radp, dp, tile_size, tw, th, packed=True, n_images=C, image_ids=ci,
                                   gaussian_ids=gi, conics=con, opacities=op)
    else:
        _, ids, fl = G.isect_tiles(m2, rad, d, tile_size, tw, th, conics=con, opacities=op)
    off = G.isect_offset_encode(ids, C, tw, th)
    leaves = [t.clone().requires_grad_(True) for t in (m2, con, colors, op)]
    bgl = backgrounds.clone().requires_grad_(True) if bg else None
    rc, ra = G.rasterize_to_pixels(leaves[0], leaves[1], leaves[2], leaves[3], W, H, tile_size, off, fl,
                                   backgrounds=bgl, masks=tile_masks, packed=packed, absgrad=absgrad)
    rc_o, ra_o, li_o = O.rasterize_to_pixels(cpu(m2), cpu(con), cpu(colors), cpu(op), W, H, tile_size, cpu(off),
                                             cpu(fl), backgrounds=cpu(backgrounds), masks=cpu(tile_masks))
    # hardware exp vs libm exp can flip a threshold decision on a few pixels: allow 1e-4 of them
    assert_close_ratio(cpu(rc), rc_o, 1e-4, 2e-5, max_bad_ratio=1e-4, name="render_colors")
    assert_close_ratio(cpu(ra), ra_o, 1e-4, 2e-5, max_bad_ratio=1e-4, name="render_alphas")
    v_rc, v_ra = torch.randn(rc_o.shape, generator=g), torch.randn(ra_o.shape, generator=g)
    ((rc * v_rc.to(DEV)).sum() + (ra * v_ra.to(DEV)).sum()).backward()
    # the oracle backward starts from ITS OWN forward state, like the kernel does from its own
    go = O.rasterize_to_pixels_bwd(cpu(m2), cpu(con), cpu(colors), cpu(op), W, H, tile_size, cpu(off), cpu(fl), ra_o,
                                   li_o, v_rc, v_ra, backgrounds=cpu(backgrounds), masks=cpu(tile_masks),
                                   absgrad=absgrad)
    for leaf, key in zip(leaves, ("v_means2d", "v_conics", "v_colors", "v_opacities")):
        assert_grad_close(cpu(leaf.grad), go[key].reshape(leaf.shape), rel=2e-3, max_bad_ratio=2e-4, name=key)
    if bg:
        assert_grad_close(cpu(bgl.grad), go["v_backgrounds"], rel=1e-3, name="v_backgrounds")
    if absgrad:
        assert_grad_close(cpu(leaves[0].absgrad), go["v_means2d_abs"].reshape(leaves[0].shape), rel=2e-3,
                          max_bad_ratio=2e-4, name="absgrad")
    return rc, ra


@pytest.mark.parametrize("D", [1, 3, 5, 32, 40])
def test_rasterize_channels(G, O, D):
    _raster_case(G, O, N=6000, C=2, W=200, H=136, tile_size=16, D=D, seed=10 + D, bg=True)


@pytest.mark.parametrize("D,absgrad", [(1, True), (2, True), (4, True), (12, True), (5, False), (8, False), (9, False), (7, True)])
def test_rasterize_one_wave_backward_absgrad_and_wide(G, O, D, absgrad):
    """The one-wave-per-tile backward (csrc/raster3d_bwd.hip, variant W) beyond 3 channels without absgrad: |v_means2d| formed
    per pixel in the turn (1 - 4 channels), and five to eight channels taken four per launch (every launch adds its share of
    the geometry gradients; not with absgrad, which is not linear in the channels: those cases and the wider ones run the
    reduction kernel)."""
    _raster_case(G, O, N=5000, C=1, W=176, H=120, tile_size=16, D=D, seed=70 + D, bg=(D % 2 == 0), absgrad=absgrad)


@pytest.mark.parametrize("D,packed,masks", [(17, False, True), (20, True, False), (31, False, False), (70, False, True), (6, True, True)])
def test_rasterize_wide_channels_on_the_matrix_cores(G, O, D, packed, masks):
    """csrc/raster3d_{fwd,bwd}_m.hip (fp32 MFMA): channel counts that are not multiples of 4 or 16, one and two column blocks,
    more than 32 channels in chunks (70 = 32 + 32 + 6: the last chunk runs the four-wave forward), tile masks, packed rows,
    an image that is not a multiple of the tile, two images."""
    _raster_case(G, O, N=5000, C=2, W=168, H=120, tile_size=16, D=D, seed=90 + D, bg=(D != 31), masks=masks, packed=packed)


@pytest.mark.parametrize("tile_size", [16, 8, 4])
def test_rasterize_tile_sizes(G, O, tile_size):
    _raster_case(G, O, N=3000, C=1, W=150, H=100, tile_size=tile_size, D=3, seed=20 + tile_size, bg=True)


def test_rasterize_masks_absgrad_packed(G, O):
    _raster_case(G, O, N=6000, C=2, W=200, H=136, tile_size=16, D=3, seed=31, bg=True, masks=True, absgrad=True)
    _raster_case(G, O, N=6000, C=2, W=200, H=136, tile_size=16, D=3, seed=32, bg=False, packed=True)


def test_rasterize_dense_overdraw_long_lists(G, O):
    """Many large splats: per-tile lists span several 256-entry batches and pixels saturate (early exit)."""
    sc, W, H = make_scene(N=8000, C=1, width=96, height=64, seed=40, scale_range=(0.3, 0.8))
    sc["opacities"][:] = 0.9
    a, rad, m2, d, con, op = _project_scene(G, sc, W, H)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    _, ids, fl = G.isect_tiles(m2, rad, d, 16, tw, th, conics=con, opacities=op)
    off = G.isect_offset_encode(ids, 1, tw, th)
    assert (torch.diff(torch.cat([off.flatten(), torch.tensor([ids.numel()], device=DEV, dtype=torch.int32)])).max()
            > 1000)
    colors = torch.rand(1, 8000, 3, device=DEV)
    rc, ra = G.rasterize_to_pixels(m2, con, colors, op, W, H, 16, off, fl)
    rc_o, ra_o, _ = O.rasterize_to_pixels(cpu(m2), cpu(con), cpu(colors), cpu(op), W, H, 16, cpu(off), cpu(fl))
    assert_close_ratio(cpu(rc), rc_o, 1e-4, 5e-5, max_bad_ratio=1e-3, name="render_colors")
    assert_close_ratio(cpu(ra), ra_o, 1e-4, 5e-5, max_bad_ratio=1e-3, name="render_alphas")
    assert ra.max() <= 1.0 and ra.min() >= 0.0 and (ra > 0.999).float().mean() > 0.5


@pytest.mark.parametrize("kind", ["expanded-scalar", "channel-slice", "row-broadcast"])
def test_rasterize_bwd_reads_cotangent_views_in_place(G, kind):
    """autograd hands v_render_colors over as a VIEW (the gradient of sum() is one float expanded to [I, H, W, D]; a slice of a
    wider image; ...). Layouts that are linear in the pixel index are read in place by the per-tile launches
    (gsx_raster3d_bwd_ws, v_colors_pixel_stride); the gradients must be those of the materialised copy."""
    from gsplat_amd import _ops
    sc, W, H = make_scene(N=5000, C=2, width=176, height=120, seed=77)
    a, rad, m2, d, con, op = _project_scene(G, sc, W, H)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    _, ids, fl = G.isect_tiles(m2, rad, d, 16, tw, th, conics=con, opacities=op)
    off = G.isect_offset_encode(ids, 2, tw, th)
    colors = torch.rand(2, 5000, 3, device=DEV)
    rc, ra, _, last = _ops.rasterize_to_pixels_3dgs(m2, con, colors, op, None, None, W, H, 16, off, fl, False, False)
    if kind == "expanded-scalar":
        v = torch.tensor(0.75, device=DEV).expand(rc.shape)
    elif kind == "channel-slice":
        v = torch.randn(2, H, W, 5, device=DEV)[..., 1:4]
    else:  # one value per channel, broadcast over the pixels
        v = torch.randn(3, device=DEV).expand(rc.shape)
    assert not v.is_contiguous() and _ops._pixel_linear_strides(v) is not None
    args = (m2, con, colors, op, None, None, off, fl, ra, last, W, H, 16, False)
    got = _ops.rasterize_to_pixels_3dgs_bwd(*args, v, None, False)
    want = _ops.rasterize_to_pixels_3dgs_bwd(*args, v.contiguous(), None, False)
    for g_, w_, name in zip(got[1:5], want[1:5], ("v_means2d", "v_conics", "v_colors", "v_opacities")):
        assert_grad_close(cpu(g_), cpu(w_), rel=1e-4, max_bad_ratio=1e-4, name=name)
    # a layout that is NOT linear in the pixel index is copied, not misread
    t = torch.randn(2, W, H, 3, device=DEV).transpose(1, 2)
    assert _ops._pixel_linear_strides(t) is None
    got = _ops.rasterize_to_pixels_3dgs_bwd(*args, t, None, False)
    want = _ops.rasterize_to_pixels_3dgs_bwd(*args, t.contiguous(), None, False)
    assert_grad_close(cpu(got[3]), cpu(want[3]), rel=1e-4, max_bad_ratio=1e-4, name="v_colors (copied)")


def test_golden_rasterize_vs_reference_outputs(G, golden):
    """HIP kernels against outputs of the REFERENCE's accumulate()+autograd (committed fixture)."""
    W, H, ts = (int(v) for v in golden["rast_wh"])
    t = {k: to_t(golden["rast_" + k], DEV) for k in ("means2d", "conics", "colors", "opacities", "offsets",
                                                     "flatten_ids", "backgrounds")}
    leaves = [t[k].clone().requires_grad_(True) for k in ("means2d", "conics", "colors", "opacities", "backgrounds")]
    rc, ra = G.rasterize_to_pixels(leaves[0], leaves[1], leaves[2], leaves[3], W, H, ts, t["offsets"],
                                   t["flatten_ids"], backgrounds=leaves[4])
    assert_close_ratio(cpu(rc), golden["rast_render_colors"], 1e-4, 2e-5, max_bad_ratio=1e-4, name="render_colors")
    assert_close_ratio(cpu(ra), golden["rast_render_alphas"], 1e-4, 2e-5, max_bad_ratio=1e-4, name="render_alphas")
    ((rc * to_t(golden["rast_v_render_colors"], DEV)).sum() + (ra * to_t(golden["rast_v_render_alphas"], DEV)).sum()).backward()
    # the reference's own PER-ELEMENT band for this op, on top of the scale-relative + cosine check (which alone would let
    # small-magnitude rows be arbitrarily wrong); the allowed share of outliers is twice what two fp32 CPU evaluations of the
    # same fixture show against each other (tests/_util.py)
    from _util import RASTER_BWD_BAND, RASTER_BWD_BAND_CPU_ENVELOPE

    for leaf, key in zip(leaves, ("v_means2d", "v_conics", "v_colors", "v_opacities", "v_backgrounds")):
        assert_grad_close(cpu(leaf.grad), golden["rast_" + key], rel=2e-3, max_bad_ratio=2e-4, name=key)
        rtol, atol = RASTER_BWD_BAND[key]
        assert torch.isfinite(leaf.grad).all(), key
        assert_close_ratio(cpu(leaf.grad), golden["rast_" + key], rtol, atol,
                           max_bad_ratio=max(2 * RASTER_BWD_BAND_CPU_ENVELOPE[key], 1e-4), name=key + " per element")


def test_golden_isect_and_projection_vs_reference_outputs(G, golden):
    ts, tw, th = (int(v) for v in golden["isect_tile"])
    m2, rad, dep = (to_t(golden[k], DEV) for k in ("isect_means2d", "isect_radii", "isect_depths"))
    tpg, ids, fl = G.isect_tiles(m2, rad, dep, ts, tw, th)
    assert torch.equal(cpu(tpg), to_t(golden["isect_tiles_per_gauss"]))
    assert torch.equal(cpu(ids), to_t(golden["isect_ids"])) and torch.equal(cpu(fl), to_t(golden["isect_flatten_ids"]))
    assert torch.equal(cpu(G.isect_offset_encode(ids, m2.shape[0], tw, th)), to_t(golden["isect_offsets"]))
    W, H = (int(v) for v in golden["proj_wh"])
    for cam in ("pinhole", "ortho", "fisheye"):
        rad, m2, d, con, comp = G.fully_fused_projection(
            to_t(golden["proj_means"], DEV), None, to_t(golden["proj_quats"], DEV), to_t(golden["proj_scales"], DEV),
            to_t(golden["proj_viewmats"], DEV), to_t(golden["proj_Ks"], DEV), W, H, calc_compensations=True,
            camera_model=cam)
        r_ref = to_t(golden[f"proj_{cam}_radii"])
        valid = (cpu(rad) > 0).all(-1) & (r_ref > 0).all(-1)
        assert ((cpu(rad) > 0).all(-1) == (r_ref > 0).all(-1)).float().mean() > 0.999
        assert (cpu(rad)[valid] - r_ref[valid]).abs().max() <= 1
        assert_close_ratio(cpu(m2)[valid], to_t(golden[f"proj_{cam}_means2d"])[valid], 1e-4, 1e-4, name="means2d")
        assert_close_ratio(cpu(con)[valid], to_t(golden[f"proj_{cam}_conics"])[valid], 1e-4, 1e-4, name="conics")
        assert_close_ratio(cpu(comp)[valid], to_t(golden[f"proj_{cam}_comps"])[valid], 1e-4, 1e-3, name="comps")
    for deg in range(5):
        col = G.spherical_harmonics(deg, to_t(golden["proj_means"], DEV), to_t(golden["proj_viewmats"], DEV),
                                    to_t(golden["sh_coeffs"], DEV))
        assert_close_ratio(cpu(col), golden[f"sh_colors_deg{deg}"], 1e-5, 1e-5, name=f"sh{deg}")


@pytest.mark.parametrize("case", ["small", "ties", "long_tiles", "many_images", "huge_tile"])
def test_tile_sort_matches_stable_sort(G, case):
    """gsx_isect_tile_sort (bucket by tile + per-tile LDS depth sort) must equal a STABLE ascending sort of the full
    keys: exact isect_ids and flatten_ids, including exact depth ties (-> ascending flatten id), tiles longer than the
    small / large LDS capacities and the global-memory path."""
    from gsplat_amd import _cabi
    from gsplat_amd._cabi import call, ptr

    g = torch.Generator().manual_seed(7)
    I, tw, th = 1, 12, 7
    n = 50_000
    if case == "many_images":
        I, tw, th, n = 3, 40, 30, 200_000
    tiles = torch.randint(0, I * tw * th, (n,), generator=g)
    if case == "long_tiles":
        tiles = torch.where(torch.rand(n, generator=g) < 0.5, torch.randint(0, 4, (n,), generator=g), tiles)  # ~6k / tile
    if case == "huge_tile":
        tiles = torch.where(torch.rand(n, generator=g) < 0.6, torch.tensor(5), tiles)  # ~30k in one tile: global path
    if case == "small":
        n = 300
        tiles = tiles[:n]
    depth = torch.rand(n, generator=g) * 10 + 0.1
    if case in ("ties", "long_tiles"):
        depth = torch.round(depth * 20) / 20  # many exact ties
    n_tiles = tw * th
    tile_bits = (n_tiles - 1).bit_length()
    img, t = tiles // n_tiles, tiles % n_tiles
    keys = (((img << tile_bits) | t) << 32) | depth.float().view(torch.int32).long()
    vals = torch.arange(n, dtype=torch.int32)  # emission order = ascending flatten id
    # shuffle rows that belong to different tiles only through the natural interleaving (keep id ascending overall)
    order = torch.sort(keys, stable=True).indices
    exp_k, exp_v = keys[order], vals[order]
    assert _cabi.tile_sort_supported(I, tw, th)
    kd, vd = keys.to(DEV), vals.to(DEV)
    ko, vo = torch.empty_like(kd), torch.empty_like(vd)
    ws = torch.empty(_cabi.tile_sort_workspace_bytes(n, I, tw, th), device=DEV, dtype=torch.uint8)
    call("gsx_isect_tile_sort", ptr(kd), ptr(vd), n, I, tw, th, ptr(ko), ptr(vo), ptr(ws), ws.numel())
    assert torch.equal(cpu(ko), exp_k), "sorted keys differ"
    assert torch.equal(cpu(vo), exp_v), "sorted flatten ids differ (tie order)"
    assert torch.equal(cpu(kd), keys) and torch.equal(cpu(vd), vals)  # inputs untouched


@pytest.mark.parametrize("cam", ["pinhole", "ortho", "fisheye"])
def test_proj_simple_fwd_bwd(G, O, cam):
    """gsplat.proj (projection_ewa_simple) vs the oracle restatement of _persp_proj/_ortho_proj/_fisheye_proj,
    gradients (including an asymmetric v_covars2d) vs torch autograd; reference test: tests/test_basic.py test_proj."""
    g = torch.Generator().manual_seed(21)
    C, N, W, H = 2, 3000, 640, 480
    means = torch.randn(C, N, 3, generator=g) * 0.8
    means[..., 2] = means[..., 2].abs() + 0.5
    A = torch.randn(C, N, 3, 3, generator=g) * 0.2
    covars = A @ A.transpose(-1, -2) + 1e-3 * torch.eye(3)
    Ks = torch.tensor([[300.0, 0, 320], [0, 310.0, 240], [0, 0, 1]]).repeat(C, 1, 1)
    lo = [means.clone().requires_grad_(True), covars.clone().requires_grad_(True)]
    lg = [means.to(DEV).requires_grad_(True), covars.to(DEV).requires_grad_(True)]
    m_o, c_o = O.proj(lo[0], lo[1], Ks, W, H, cam)
    m_g, c_g = G.proj(lg[0], lg[1], Ks.to(DEV), W, H, cam)
    assert_close_ratio(cpu(m_g), m_o, 1e-4, 1e-3, name="means2d")
    assert_grad_close(cpu(c_g), c_o.detach(), rel=1e-5, name="covars2d")
    v_m, v_c = torch.randn(m_o.shape, generator=g), torch.randn(c_o.shape, generator=g) * 1e-3
    ((m_o * v_m).sum() + (c_o * v_c).sum()).backward()
    ((m_g * v_m.to(DEV)).sum() + (c_g * v_c.to(DEV)).sum()).backward()
    assert_grad_close(cpu(lg[0].grad), lo[0].grad, rel=2e-3, max_bad_ratio=2e-4, name="v_means")
    assert_grad_close(cpu(lg[1].grad), lo[1].grad, rel=1e-4, name="v_covars")


@pytest.mark.parametrize("layout", ["dense", "dense-masked", "packed"])
@pytest.mark.parametrize("deg,K", [(3, 16), (2, 16), (1, 4), (4, 25)])
def test_split_sh_matches_oracle(G, O, layout, deg, K):
    """spherical_harmonics_l0 + spherical_harmonics_l1_plus (shN [N, K-1, 3] read in place: csrc/sh_band.hip) against the
    ORACLE's full evaluation on the concatenated coefficients (reference tests/test_basic.py split-SH variants;
    SphericalHarmonicsL1PlusCUDA.cu:441, 648): colours, v_sh0, v_shN, v_means, and the pose gradient."""
    N, C = 3011, 3  # not a multiple of 64: the last wave's block is ragged
    sc, W, H = make_scene(N=N, C=C, seed=6)
    g = torch.Generator().manual_seed(deg * 100 + K)
    coeffs = torch.randn(N, K, 3, generator=g) * 0.3
    masks = (torch.rand(C, N, generator=g) > 0.3) if layout != "dense" else None
    d = {k: v.to(DEV) for k, v in sc.items()}
    sh0 = coeffs[:, :1].to(DEV).requires_grad_(True)
    shN = coeffs[:, 1:].contiguous().to(DEV).requires_grad_(True)
    mg = d["means"].clone().requires_grad_(True)
    vg = d["viewmats"].clone().requires_grad_(True)
    if layout == "packed":
        ci, gi = torch.where(masks)
        bi = torch.zeros_like(ci)
        # the reference's packed contract: coefficient rows pre-gathered to [nnz, K - 1, 3] (SphericalHarmonics.cpp:90-104)
        got = G.spherical_harmonics_l0(sh0)[gi.to(DEV)] + G.spherical_harmonics_l1_plus(
            deg, mg, vg, shN[gi.to(DEV)], batch_ids=bi.to(DEV), camera_ids=ci.to(DEV), gaussian_ids=gi.to(DEV))
    else:
        got = G.spherical_harmonics_l1_plus(deg, mg, vg, shN, masks=None if masks is None else masks.to(DEV))
        l0 = G.spherical_harmonics_l0(sh0)[None].expand_as(got)
        got = got + (l0 if masks is None else l0 * masks.to(DEV)[..., None])
    co = coeffs.clone().requires_grad_(True)
    mo, vo = sc["means"].clone().requires_grad_(True), sc["viewmats"].clone().requires_grad_(True)
    ref = O.spherical_harmonics(deg, mo[None], vo[None], co, None if masks is None else masks[None])[0]
    if layout == "packed":
        ref = ref[masks]
    assert_close_ratio(cpu(got), ref, 1e-5, 1e-5, name="split sh colours")
    w = torch.randn(ref.shape, generator=g)
    (got * w.to(DEV)).sum().backward()
    (ref * w).sum().backward()
    assert_grad_close(cpu(torch.cat([sh0.grad, shN.grad], 1)), co.grad, rel=1e-5, name="v_sh0 | v_shN")
    assert_grad_close(cpu(mg.grad), mo.grad, rel=1e-4, name="v_means")
    assert_grad_close(cpu(vg.grad), vo.grad, rel=2e-4, name="v_viewmats")


@pytest.mark.parametrize("split", [False, True])
def test_sh_half_coefficients(G, O, split):
    """fp16 coefficient rows (reference SphericalHarmonicsCUDA.cu:609-638 / SphericalHarmonicsL1PlusCUDA.cu:569): read in
    place as halves, float arithmetic and colours; the oracle evaluates the SAME (exactly widened) coefficients, so the
    colours agree to fp32 tolerance; the coefficient gradient comes back in half precision."""
    N, C, deg, K = 2500, 2, 3, 16
    sc, W, H = make_scene(N=N, C=C, seed=8)
    g = torch.Generator().manual_seed(5)
    coeffs = (torch.randn(N, K, 3, generator=g) * 0.3).half()
    masks = torch.rand(C, N, generator=g) > 0.2
    d = {k: v.to(DEV) for k, v in sc.items()}
    mg = d["means"].clone().requires_grad_(True)
    if split:
        shN = coeffs[:, 1:].contiguous().to(DEV).requires_grad_(True)
        got = G.spherical_harmonics_l1_plus(deg, mg, d["viewmats"], shN, masks=masks.to(DEV))
        co = torch.cat([torch.zeros(N, 1, 3), coeffs[:, 1:].float()], 1).requires_grad_(True)
    else:
        cg = coeffs.to(DEV).requires_grad_(True)
        got = G.spherical_harmonics(deg, mg, d["viewmats"], cg, masks=masks.to(DEV))
        co = coeffs.float().requires_grad_(True)
    assert got.dtype == torch.float32
    mo = sc["means"].clone().requires_grad_(True)
    ref = O.spherical_harmonics(deg, mo[None], sc["viewmats"][None], co, masks[None])[0]
    assert_close_ratio(cpu(got), ref, 1e-5, 1e-5, name="half-coefficient colours")
    w = torch.randn(ref.shape, generator=g)
    (got * w.to(DEV)).sum().backward()
    (ref * w).sum().backward()
    v = shN.grad if split else cg.grad
    assert v.dtype == torch.float16
    want = co.grad[:, 1:] if split else co.grad
    assert_grad_close(cpu(v).float(), want, rel=2e-3, name="v_coeffs (half)")
    assert_grad_close(cpu(mg.grad), mo.grad, rel=1e-4, name="v_means")


def test_rasterize_to_indices_matches_oracle(G, O):
    """rasterize_to_indices_in_range{,_2dgs}: contributing (gaussian, pixel, image) triples in the reference's order
    (pixel-major, list order). Full range vs the oracle; a split range chained through the transmittance must give the
    same set (the reference's torch rasterizer relies on that: _torch_impl.py:871-905)."""
    sc, W, H = make_scene(N=1500, C=2, width=72, height=56, seed=8)
    a, rad, m2, d, con, op = _project_scene(G, sc, W, H)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    _, ids, fl = G.isect_tiles(m2, rad, d, 16, tw, th, conics=con, opacities=op)
    off = G.isect_offset_encode(ids, 2, tw, th)
    T0 = torch.ones(2, H, W, device=DEV)
    gi, pi, ii = G.rasterize_to_indices_in_range(0, 10 ** 6, T0, m2, con, op, W, H, 16, off, fl)
    go, po, io = O.rasterize_to_indices(cpu(m2), cpu(con), cpu(op), W, H, 16, cpu(off), cpu(fl))
    # oracle order is tile-major; compare as sorted (image, pixel)-grouped sequences keeping list order within a pixel
    def canon(g_, p_, i_):
        key = i_ * (W * H) + p_
        order = torch.sort(key, stable=True).indices
        return torch.stack([key[order], g_[order]], -1)
    ca, cb = canon(cpu(gi), cpu(pi), cpu(ii)), canon(go, po, io)
    assert ca.shape == cb.shape
    assert (ca != cb).any(-1).float().mean() < 1e-4  # exp rounding may flip a threshold on a handful of pairs
    assert torch.equal(torch.sort(cpu(ii) * (W * H) + cpu(pi)).values, cpu(ii) * (W * H) + cpu(pi))  # pixel-major order
    # 2DGS variant
    dsc = {k: v.to(DEV) for k, v in sc.items()}
    rad2, m22, d2, M2, _ = G.fully_fused_projection_2dgs(dsc["means"], dsc["quats"], dsc["scales"], dsc["viewmats"],
                                                         dsc["Ks"], W, H)
    _, ids2, fl2 = G.isect_tiles(m22, rad2, d2, 16, tw, th)
    off2 = G.isect_offset_encode(ids2, 2, tw, th)
    g2, p2, i2 = G.rasterize_to_indices_in_range_2dgs(0, 10 ** 6, T0, m22, M2, op, W, H, 16, off2, fl2)
    g2o, p2o, i2o = O.rasterize_to_indices_2dgs(cpu(m22), cpu(M2), cpu(op), W, H, 16, cpu(off2), cpu(fl2))
    ca, cb = canon(cpu(g2), cpu(p2), cpu(i2)), canon(g2o, p2o, i2o)
    assert ca.shape == cb.shape and (ca != cb).any(-1).float().mean() < 1e-4


# ------------------------------------------------------------------------------------------------
# query rasterizers (reference tests/test_basic.py contributing-ids tests compare against the
# same front-to-back walk)
@pytest.mark.parametrize("packed", [False, True])
def test_query_rasterizers_match_oracle(G, O, packed):
    sc, W, H = make_scene(N=1500, C=2, width=72, height=56, seed=31)
    a, rad, m2, d, con, op = _project_scene(G, sc, W, H)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    C, N = m2.shape[0], m2.shape[1]
    if packed:
        vis = (rad > 0).all(-1)
        ci, gi = torch.where(vis)
        rows = ci * N + gi
        m2p, conp, opp, radp, dp = (t.reshape(C * N, -1)[rows].squeeze(-1) if t.dim() == 2 else t.reshape(C * N, -1)[rows]
                                    for t in (m2, con, op, rad, d))
        opp, dp = opp.reshape(-1), dp.reshape(-1)
        _, ids_s, fl = G.isect_tiles(m2p, radp, dp, 16, tw, th, packed=True, n_images=C, image_ids=ci, gaussian_ids=gi)
        off = G.isect_offset_encode(ids_s, C, tw, th)
        args = (m2p, conp, opp, off, fl, W, H, 16)
    else:
        _, ids_s, fl = G.isect_tiles(m2, rad, d, 16, tw, th)
        off = G.isect_offset_encode(ids_s, C, tw, th)
        args = (m2, con, op, off, fl, W, H, 16)
    # oracle on the dense layout (same Gaussians; packed ids are rows of the packed arrays -> map back for comparison)
    _, ids_o, fl_o = O.isect_tiles(cpu(m2), cpu(rad), cpu(d), 16, tw, th)
    off_o = O.isect_offset_encode(ids_o, C, tw, th)
    lids, lwts, alphas_o = O.raster_contributions(cpu(m2), cpu(con), cpu(op), W, H, 16, off_o, fl_o)
    counts_o = torch.tensor([len(x) for x in lids], dtype=torch.int32).reshape(C, H, W)

    counts, alphas = G.rasterize_num_contributing_gaussians(*args)
    assert counts.dtype == torch.int32 and counts.shape == (C, H, W)
    bad = (cpu(counts) != counts_o).float().mean().item()
    assert bad <= 2e-3, f"per-pixel contributor counts differ on {bad:.4%} of the pixels"
    assert_close_ratio(cpu(alphas), alphas_o, 1e-4, 5e-5, max_bad_ratio=1e-3, name="query alphas")

    ids, wts = G.rasterize_contributing_gaussian_ids(*args, counts)
    kmax = int(counts.max())
    assert ids.shape == (C, H, W, kmax) and wts.shape == ids.shape
    ids_c, wts_c = cpu(ids).reshape(-1, kmax), cpu(wts).reshape(-1, kmax)
    if packed:  # packed ids are rows of the packed arrays: translate to Gaussian ids
        gi_c = cpu(gi)
        ids_c = torch.where(ids_c >= 0, gi_c[ids_c.clamp_min(0)], ids_c)
    same = cpu(counts).reshape(-1) == counts_o.reshape(-1)
    n_checked = 0
    for p in torch.where(same)[0][::7].tolist():  # every 7th pixel with an identical count
        k = len(lids[p])
        assert ids_c[p, :k].tolist() == lids[p], p
        assert (ids_c[p, k:] == -1).all() and (wts_c[p, k:] == 0).all()
        torch.testing.assert_close(wts_c[p, :k].double(), torch.tensor(lwts[p], dtype=torch.float64), rtol=2e-3, atol=1e-6)
        n_checked += 1
    assert n_checked > 200

    K = 4
    tids, twts = G.rasterize_top_contributing_gaussian_ids(*args, K)
    assert tids.shape == (C, H, W, K)
    tids_c, twts_c = cpu(tids).reshape(-1, K), cpu(twts).reshape(-1, K)
    if packed:
        tids_c = torch.where(tids_c >= 0, gi_c[tids_c.clamp_min(0)], tids_c)
    oi, ow = O.top_contributions(lids, lwts, K)
    pix = torch.where(same)[0]
    id_match = (tids_c[pix] == torch.from_numpy(oi)[pix]).all(-1).float().mean().item()
    assert id_match >= 0.995, f"top-{K} ids agree on only {id_match:.4%} of the pixels"  # near-ties in alpha*T may swap
    agree = pix[(tids_c[pix] == torch.from_numpy(oi)[pix]).all(-1)]
    torch.testing.assert_close(twts_c[agree].double(), torch.from_numpy(ow)[agree], rtol=2e-3, atol=1e-6)
    # the kept samples are in front-to-back order: their positions in the full list increase
    for p in agree[::97].tolist():
        pos = [lids[p].index(g) for g in tids_c[p].tolist() if g >= 0]
        assert pos == sorted(pos)


@pytest.mark.parametrize("batch_dims", [(), (2,)])
@pytest.mark.parametrize("via", ["wrapper", "op"])
def test_isect_tiles_float64_rows(G, O, batch_dims, via):
    """float64 rows (the reference instantiates intersect_tile for double, tests/test_basic.py:1268-1316): bit-exact against
    the oracle's double branch (itself pinned against the reference's torch restatement), dense with batch dimensions and
    packed, through isect_tiles() and through the raw op (the compiled body); the exact test stays fp32 (TypeError)."""
    g = torch.Generator().manual_seed(42)
    C, N, width, height, ts = 3, 1000, 40, 60, 16
    shape = tuple(batch_dims) + (C, N)
    means2d = torch.randn(shape + (2,), generator=g, dtype=torch.float64) * width
    radii = torch.randint(0, width, shape + (2,), generator=g, dtype=torch.int32)
    depths = torch.rand(shape, generator=g, dtype=torch.float64)
    tw, th = math.ceil(width / ts), math.ceil(height / ts)
    I = math.prod(batch_dims) * C
    want = O.isect_tiles(means2d, radii, depths, ts, tw, th)
    if via == "wrapper":
        got = G.isect_tiles(means2d.to(DEV), radii.to(DEV), depths.to(DEV), ts, tw, th)
    else:
        got = torch.ops.gsplat.intersect_tile(means2d.to(DEV), radii.to(DEV), depths.to(DEV), None, None, None, None, None, ts,
                                              tw, th, True, False)
    for a, b, nm in zip(got, want, ("tiles_per_gauss", "isect_ids", "flatten_ids")):
        assert a.dtype == b.dtype and torch.equal(cpu(a), b), nm
    assert torch.equal(cpu(G.isect_offset_encode(got[1], I, tw, th)).reshape(-1), O.isect_offset_encode(want[1], I, tw, th).reshape(-1))
    # packed rows of several images
    flat = lambda t, k: t.reshape((-1,) + t.shape[len(shape):]) if k else t.reshape(-1)
    keep = torch.rand(I * N, generator=g) > 0.4
    image_ids = (torch.arange(I * N) // N)[keep]
    m_p, r_p, d_p = flat(means2d, 1)[keep], flat(radii, 1)[keep], flat(depths, 0)[keep]
    want_p = O.isect_tiles(m_p, r_p, d_p, ts, tw, th, image_ids=image_ids, n_images=I)
    got_p = G.isect_tiles(m_p.to(DEV), r_p.to(DEV), d_p.to(DEV), ts, tw, th, packed=True, n_images=I,
                          image_ids=image_ids.to(DEV), gaussian_ids=torch.zeros_like(image_ids).to(DEV))
    for a, b in zip(got_p, want_p):
        assert torch.equal(cpu(a), b)
    with pytest.raises(RuntimeError):  # a failed check of a dispatcher op is a RuntimeError, like the reference's TORCH_CHECK
        torch.ops.gsplat.intersect_tile(means2d.to(DEV), radii.to(DEV), depths.to(DEV),
                                        torch.ones(shape + (3,), dtype=torch.float64, device=DEV),
                                        torch.ones(shape, dtype=torch.float64, device=DEV), None, None, None, ts, tw, th, True, False)
