"""CPU: the oracle (oracle/) against the golden vectors produced by the REFERENCE's own Python
(tests/golden/garden_quarter.npz, written by oracle/pin_against_reference.py from
gsplat/cuda/_torch_impl.py + accumulate()). Keeps the oracle pinned on machines without /root/reference."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from _util import assert_close_ratio, assert_grad_close, to_t


def test_quat_scale_to_covar_preci(golden):
    q, s = to_t(golden["qs_quats"]), to_t(golden["qs_scales"])
    c, p = O.quat_scale_to_covar_preci(q, s, True, True, True)
    assert_close_ratio(c, golden["qs_covars_triu"], 1e-5, 1e-7, name="covars")
    assert_close_ratio(p, golden["qs_precis_triu"], 1e-4, 1e-2, name="precis")


def _proj(golden, cam, requires_grad=False):
    means, quats, scales = (to_t(golden[k]).clone() for k in ("proj_means", "proj_quats", "proj_scales"))
    viewmats, Ks = to_t(golden["proj_viewmats"]).clone(), to_t(golden["proj_Ks"])
    W, H = (int(v) for v in golden["proj_wh"])
    if requires_grad:
        for t in (means, quats, scales, viewmats):
            t.requires_grad_(True)
    out = O.fully_fused_projection(means[None], None, quats[None], scales[None], viewmats[None], Ks[None], W, H, 0.3,
                                   0.01, 1e10, 0.0, True, cam, None)
    return (means, quats, scales, viewmats), [o[0] for o in out]


def test_projection_forward_all_camera_models(golden):
    for cam in ("pinhole", "ortho", "fisheye"):
        _, (radii, m2, d, con, comp) = _proj(golden, cam)
        r_ref = to_t(golden[f"proj_{cam}_radii"])
        valid = (radii > 0).all(-1) & (r_ref > 0).all(-1)
        assert ((radii > 0).all(-1) == (r_ref > 0).all(-1)).float().mean() > 0.999
        assert (radii[valid] - r_ref[valid]).abs().max() <= 1
        assert_close_ratio(m2[valid], to_t(golden[f"proj_{cam}_means2d"])[valid], 1e-4, 1e-4, name=f"{cam} means2d")
        assert_close_ratio(d[valid], to_t(golden[f"proj_{cam}_depths"])[valid], 1e-4, 1e-4, name=f"{cam} depths")
        assert_close_ratio(con[valid], to_t(golden[f"proj_{cam}_conics"])[valid], 1e-4, 1e-4, name=f"{cam} conics")
        assert_close_ratio(comp[valid], to_t(golden[f"proj_{cam}_comps"])[valid], 1e-4, 1e-3, name=f"{cam} comps")


def test_projection_backward_matches_reference_autograd(golden):
    for cam in ("pinhole", "ortho", "fisheye"):
        inputs, (radii, m2, d, con, comp) = _proj(golden, cam, requires_grad=True)
        w = to_t(golden[f"proj_{cam}_w"])
        valid = to_t(golden[f"proj_{cam}_valid"])
        vm = valid[..., None].float()
        loss = ((m2 * w[..., 0:2] * vm).sum() + (d * w[..., 2] * valid).sum() + (con * w[..., 3:6] * vm).sum()
                + (comp * w[..., 6] * valid).sum())
        grads = torch.autograd.grad(loss, inputs)
        for nm, g in zip(("v_means", "v_quats", "v_scales", "v_viewmats"), grads):
            assert_grad_close(g, golden[f"proj_{cam}_{nm}"], rel=2e-3, name=f"{cam} {nm}")


def test_spherical_harmonics(golden):
    means, viewmats = to_t(golden["proj_means"]), to_t(golden["proj_viewmats"])
    coeffs = to_t(golden["sh_coeffs"])
    for deg in range(5):
        c = O.spherical_harmonics(deg, means[None], viewmats[None], coeffs)[0]
        assert_close_ratio(c, golden[f"sh_colors_deg{deg}"], 1e-5, 1e-5, name=f"sh deg {deg}")


def test_isect_exact(golden):
    ts, tw, th = (int(v) for v in golden["isect_tile"])
    m2, rad, dep = to_t(golden["isect_means2d"]), to_t(golden["isect_radii"]), to_t(golden["isect_depths"])
    tpg, ids, fl = O.isect_tiles(m2, rad, dep, ts, tw, th, sort=True)
    assert torch.equal(tpg, to_t(golden["isect_tiles_per_gauss"]))
    assert torch.equal(ids, to_t(golden["isect_ids"]))
    assert torch.equal(fl, to_t(golden["isect_flatten_ids"]))
    off = O.isect_offset_encode(ids, m2.shape[0], tw, th)
    assert torch.equal(off, to_t(golden["isect_offsets"]))
    # ellipse mode: committed oracle vector (no Python restatement exists in the reference)
    tpa, ida, fla = O.isect_tiles(m2, rad, dep, ts, tw, th, sort=True, conics=to_t(golden["isect_conics"]),
                                  opacities=to_t(golden["isect_opacities"]))
    assert torch.equal(tpa, to_t(golden["isect_accu_tiles_per_gauss"]))
    assert torch.equal(ida, to_t(golden["isect_accu_ids"]))
    assert torch.equal(fla, to_t(golden["isect_accu_flatten_ids"]))
    assert (tpa <= tpg).all()


def _pairs(ids, fl, tile_bits):
    """(image, tile, row) triples of an intersection list as one int64 key per entry."""
    ids = ids.numpy().astype(np.int64)
    tile = (ids >> 32) & ((1 << tile_bits) - 1)
    img = ids >> (32 + tile_bits)
    return (img << 48) | (tile << 24) | fl.numpy().astype(np.int64)  # rows < 2^24 in these fixtures


@pytest.mark.parametrize("source", ["isect", "proj_pinhole", "proj_fisheye"])
def test_accutile_drops_only_invisible_pairs_of_the_reference_aabb_set(golden, source):
    """The exact-ellipse tile walk (AccuTile: IntersectTile.cu:83-207, 288-373 - CUDA only in the reference, so the oracle's
    restatement of it cannot be replayed against reference vectors) is pinned by a PROPERTY instead: on the garden fixture its
    (tile, Gaussian) set is a subset of the radius-box set - which IS pinned to the reference's `_isect_tiles` golden above -
    and every pair it drops is invisible: max over the tile's pixel centres of min(0.99, opacity exp(-sigma)) < 1/255, the
    compositing kernels' own alpha test (RasterizeToPixels3DGSDevice.cuh:44-56). So dropping it cannot change a pixel."""
    if source == "isect":
        ts, tw, th = (int(v) for v in golden["isect_tile"])
        m2, rad, dep = (to_t(golden["isect_" + k]) for k in ("means2d", "radii", "depths"))
        con, op = to_t(golden["isect_conics"]), to_t(golden["isect_opacities"])
    else:
        W, H = (int(v) for v in golden["proj_wh"])
        ts = 16
        tw, th = (W + ts - 1) // ts, (H + ts - 1) // ts
        m2, rad, dep, con = (to_t(golden[f"{source}_{k}"]) for k in ("means2d", "radii", "depths", "conics"))
        valid = to_t(golden[f"{source}_valid"])
        rad = rad * valid[..., None].to(rad.dtype)  # the reference leaves the other fields of culled rows undefined
        m2, dep, con = (torch.nan_to_num(t * valid.reshape(valid.shape + (1,) * (t.dim() - 2)).to(t.dtype)) for t in (m2, dep, con))
        op = to_t(golden["proj_opacities"])[None].expand(m2.shape[:2]).contiguous()
    I, N = m2.shape[:2]
    tile_bits = O.bits_for_count(tw * th)
    _, ids_b, fl_b = O.isect_tiles(m2, rad, dep, ts, tw, th, sort=True)
    _, ids_a, fl_a = O.isect_tiles(m2, rad, dep, ts, tw, th, sort=True, conics=con, opacities=op)
    box, accu = _pairs(ids_b, fl_b, tile_bits), _pairs(ids_a, fl_a, tile_bits)
    assert np.unique(accu).size == accu.size and np.isin(accu, box).all(), "the ellipse walk visits a tile outside the radius box"
    dropped = box[~np.isin(box, accu)]
    assert 0 < dropped.size < box.size, "the fixture must exercise the exact test"
    row, tile = dropped & 0xFFFFFF, (dropped >> 24) & 0xFFFFFF
    m2f, conf, opf = m2.reshape(-1, 2).double().numpy(), con.reshape(-1, 3).double().numpy(), op.reshape(-1).double().numpy()
    u = np.arange(ts) + 0.5
    px = (tile % tw)[:, None, None] * ts + u[None, None, :]      # [pairs, 1, ts]
    py = (tile // tw)[:, None, None] * ts + u[None, :, None]     # [pairs, ts, 1]
    dx, dy = m2f[row, 0][:, None, None] - px, m2f[row, 1][:, None, None] - py
    A, B, C = (conf[row, k][:, None, None] for k in range(3))
    sigma = 0.5 * (A * dx * dx + C * dy * dy) + B * dx * dy
    alpha = np.minimum(0.99, opf[row][:, None, None] * np.exp(-sigma))
    alpha = np.where(sigma < 0, 0.0, alpha)                      # the kernels skip sigma < 0
    worst = alpha.reshape(dropped.size, -1).max(1)
    assert worst.max() < 1.0 / 255.0, f"a dropped (tile, Gaussian) pair is visible: alpha {worst.max():.6f} at pair {int(worst.argmax())}"


def _rast_inputs(golden):
    W, H, ts = (int(v) for v in golden["rast_wh"])
    keys = ("rast_means2d", "rast_conics", "rast_colors", "rast_opacities", "rast_offsets", "rast_flatten_ids",
            "rast_backgrounds")
    return (W, H, ts) + tuple(to_t(golden[k]) for k in keys)


def test_rasterize_forward_vs_reference_accumulate(golden):
    W, H, ts, m2, con, col, op, off, fl, bg = _rast_inputs(golden)
    rc, ra, li = O.rasterize_to_pixels(m2, con, col, op, W, H, ts, off, fl, backgrounds=bg)
    assert_close_ratio(rc, golden["rast_render_colors"], 1e-5, 2e-5, name="render_colors")
    assert_close_ratio(ra, golden["rast_render_alphas"], 1e-5, 2e-5, name="render_alphas")
    assert torch.equal(li, to_t(golden["rast_last_ids"]))


def test_rasterize_backward_vs_reference_autograd(golden):
    W, H, ts, m2, con, col, op, off, fl, bg = _rast_inputs(golden)
    _, ra, li = O.rasterize_to_pixels(m2, con, col, op, W, H, ts, off, fl, backgrounds=bg)
    g = O.rasterize_to_pixels_bwd(m2, con, col, op, W, H, ts, off, fl, ra, li, to_t(golden["rast_v_render_colors"]),
                                  to_t(golden["rast_v_render_alphas"]), backgrounds=bg)
    for k in ("v_means2d", "v_conics", "v_colors", "v_opacities", "v_backgrounds"):
        assert_grad_close(g[k].reshape(golden["rast_" + k].shape), golden["rast_" + k], rel=5e-4, name=k)


def test_rasterize_closed_form_single_gaussian():
    """One isotropic Gaussian centred on a pixel centre: alpha = min(.99, o*exp(-r^2/(2 s^2)))."""
    W = H = 16
    s2, o = 4.0, 0.8
    m2 = torch.tensor([[[8.5, 8.5]]])
    con = torch.tensor([[[1 / s2, 0.0, 1 / s2]]])
    col = torch.tensor([[[0.25, 0.5, 1.0]]])
    op = torch.tensor([[o]])
    off = torch.zeros(1, 1, 1, dtype=torch.int32)
    fl = torch.zeros(1, dtype=torch.int32)
    rc, ra, li = O.rasterize_to_pixels(m2, con, col, op, W, H, 16, off, fl)
    ys, xs = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
    r2 = (xs - 8.5) ** 2 + (ys - 8.5) ** 2
    a = torch.clamp_max(o * torch.exp(-0.5 * r2 / s2), 0.99)
    a = torch.where(a < 1 / 255, torch.zeros_like(a), a)
    assert_close_ratio(ra[0, ..., 0], a, 1e-6, 1e-6, name="alpha")
    assert_close_ratio(rc[0], a[..., None] * col[0, 0], 1e-6, 1e-6, name="color")


def test_rasterize_saturating_gaussian_is_excluded():
    """Two opaque splats then a third: T after two = 1e-4 -> the SECOND one already hits T' <= 1e-4 and is
    excluded (exclusive stop, RasterizeToPixels3DGSDevice.cuh:86-90)."""
    m2 = torch.tensor([[[0.5, 0.5]] * 3])
    con = torch.tensor([[[1e-6, 0.0, 1e-6]] * 3])
    col = torch.tensor([[[1.0], [2.0], [4.0]]])
    op = torch.tensor([[1.0, 1.0, 1.0]])
    off = torch.zeros(1, 1, 1, dtype=torch.int32)
    fl = torch.arange(3, dtype=torch.int32)
    rc, ra, li = O.rasterize_to_pixels(m2, con, col, op, 1, 1, 16, off, fl)
    assert abs(ra.item() - 0.99) < 1e-6 and abs(rc.item() - 0.99) < 1e-6 and li.item() == 0


def test_bits_for_count():
    assert [O.bits_for_count(n) for n in (0, 1, 2, 3, 4, 5, 8, 9, 8160)] == [0, 0, 1, 2, 2, 3, 3, 4, 13]


def test_isect_tiles_float64_matches_reference_torch_restatement():
    """float64 rows: the oracle's double branch against outputs of the reference's _isect_tiles / _isect_offset_encode run in
    float64 (tests/golden/isect_f64_ref.npz, oracle/pin_isect_f64_against_reference.py) - bit for bit."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "isect_f64_ref.npz"))
    for name in ("a", "b", "c"):
        C, N, width, height, ts, tw, th = (int(v) for v in g[f"{name}_cfg"])
        m, r, d = (torch.from_numpy(g[f"{name}_{k}"]) for k in ("means2d", "radii", "depths"))
        assert m.dtype == torch.float64
        tpg, ids, fl = O.isect_tiles(m, r, d, ts, tw, th)
        assert torch.equal(tpg.reshape(-1), torch.from_numpy(g[f"{name}_tpg"]).reshape(-1))
        assert torch.equal(ids, torch.from_numpy(g[f"{name}_ids"])) and torch.equal(fl, torch.from_numpy(g[f"{name}_flat"]))
        off = O.isect_offset_encode(ids, C, tw, th)
        assert torch.equal(off.reshape(-1), torch.from_numpy(g[f"{name}_offsets"]).reshape(-1))
    with pytest.raises(TypeError):
        O.isect_tiles(m, r, d, ts, tw, th, conics=torch.ones(C, N, 3), opacities=torch.ones(C, N))


def test_rasterize_bwd_per_element_band(golden):
    """The reference's per-element band (tests/test_basic.py:2664-2675) on the golden fixture, C oracle vs the stored outputs of
    the reference's torch autograd: two fp32 CPU evaluations of the same sums. The share of elements outside the band is the
    envelope the GPU kernels are held to twice over (tests/_util.py: RASTER_BWD_BAND_CPU_ENVELOPE)."""
    from _util import RASTER_BWD_BAND, RASTER_BWD_BAND_CPU_ENVELOPE

    W, H, ts, m2, con, col, op, off, fl, bg = _rast_inputs(golden)
    rc, ra, li = O.rasterize_to_pixels(m2, con, col, op, W, H, ts, off, fl, backgrounds=bg)
    g = O.rasterize_to_pixels_bwd(m2, con, col, op, W, H, ts, off, fl, ra, li, to_t(golden["rast_v_render_colors"]),
                                  to_t(golden["rast_v_render_alphas"]), backgrounds=bg)
    for k, (rtol, atol) in RASTER_BWD_BAND.items():
        assert_close_ratio(g[k].reshape(golden["rast_" + k].shape), golden["rast_" + k], rtol, atol,
                           max_bad_ratio=RASTER_BWD_BAND_CPU_ENVELOPE[k], name=k + " per element (CPU vs CPU)")


def test_rasterize_bwd_fp64_samples_agree_with_the_fp32_oracle(golden):
    """gso_raster3d_bwd_f64 (per-sample math in fp64: the reference value of tests/test_gpu_pipeline.py's per-element band at
    c3) against the fp32-sample oracle and the reference's stored autograd values on the golden fixture."""
    from _util import RASTER_BWD_BAND

    W, H, ts, m2, con, col, op, off, fl, bg = _rast_inputs(golden)
    rc, ra, li = O.rasterize_to_pixels(m2, con, col, op, W, H, ts, off, fl, backgrounds=bg)
    v_rc, v_ra = to_t(golden["rast_v_render_colors"]), to_t(golden["rast_v_render_alphas"])
    g32 = O.rasterize_to_pixels_bwd(m2, con, col, op, W, H, ts, off, fl, ra, li, v_rc, v_ra, backgrounds=bg, absgrad=True)
    g64 = O.rasterize_to_pixels_bwd(m2, con, col, op, W, H, ts, off, fl, ra, li, v_rc, v_ra, backgrounds=bg, absgrad=True,
                                    sample_f64=True)
    for k in g32:
        assert_grad_close(g32[k], g64[k], rel=1e-4, max_bad_ratio=1e-4, name=k)
    gs = O.rasterize_to_pixels_bwd(m2, con, col, op, W, H, ts, off, fl, ra, li, v_rc, v_ra, backgrounds=bg, sum_f32=True)
    for k in gs:
        assert_grad_close(gs[k], g64[k], rel=1e-4, max_bad_ratio=1e-4, name=k + " (fp32 sums)")
    for k, (rtol, atol) in RASTER_BWD_BAND.items():
        assert_close_ratio(g64[k].reshape(golden["rast_" + k].shape), golden["rast_" + k], rtol, atol, max_bad_ratio=1e-2,
                           name=k + " fp64 samples vs the reference's autograd")
