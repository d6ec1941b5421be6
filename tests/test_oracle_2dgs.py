"""CPU: the 2DGS oracle (oracle/oracle.py + gsplat_oracle.c) against (a) the committed outputs of the reference's own
Python (_torch_impl_2dgs.py, fixture written by oracle/pin_against_reference.py) and (b) an independent differentiable
fp64 restatement of the CUDA-only outputs (distortion, median depth) whose gradients come from torch autograd."""
import math
import os

import numpy as np
import pytest
import torch

from _util import assert_close_ratio, assert_grad_close, make_scene, to_t

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    return oracle


@pytest.fixture(scope="module")
def g2():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "garden_quarter_2dgs.npz")))


def test_projection_2dgs_vs_reference_outputs(O, g2):
    W, H = (int(v) for v in g2["wh"])
    leaves = [to_t(g2[k]).clone().requires_grad_(True) for k in ("means", "quats", "scales", "viewmats")]
    rad, m2, d, M, n = O.fully_fused_projection_2dgs(*leaves, to_t(g2["Ks"]), W, H)
    valid = to_t(g2["proj_valid"])
    assert ((rad > 0).all(-1) == (to_t(g2["proj_radii"]) > 0).all(-1)).float().mean() > 0.999
    assert (rad[valid] - to_t(g2["proj_radii"])[valid]).abs().max() <= 1
    assert_close_ratio(m2[valid], to_t(g2["proj_means2d"])[valid], 1e-4, 1e-3, name="means2d")
    assert_close_ratio(d[valid], to_t(g2["proj_depths"])[valid], 1e-5, 1e-5, name="depths")
    assert_close_ratio(M[valid], to_t(g2["proj_ray_transforms"])[valid], 1e-4, 1e-3, name="ray_transforms")
    assert_close_ratio(n[valid], to_t(g2["proj_normals"])[valid], 1e-5, 1e-5, name="normals")
    vm = valid.float()
    loss = ((m2 * to_t(g2["proj_w_means2d"])).sum(-1) * vm).sum() + (d * to_t(g2["proj_w_depths"]) * vm).sum() \
        + ((M * to_t(g2["proj_w_ray_transforms"])).sum((-1, -2)) * vm).sum() \
        + ((n * to_t(g2["proj_w_normals"])).sum(-1) * vm).sum()
    grads = torch.autograd.grad(loss, leaves)
    for nm, a in zip(("v_means", "v_quats", "v_scales", "v_viewmats"), grads):
        e = to_t(g2["proj_" + nm])
        if nm == "v_scales":
            a, e = a[:, :2], e[:, :2]
        assert_grad_close(a, e, rel=2e-3, name=nm)


def _rast_inputs(g2):
    W, H, ts = (int(v) for v in g2["rast_wh"])
    t = {k: to_t(g2["rast_" + k]) for k in ("means2d", "ray_transforms", "colors", "opacities", "normals", "offsets",
                                            "flatten_ids", "backgrounds")}
    return W, H, ts, t


def test_rasterize_2dgs_vs_reference_accumulate(O, g2):
    W, H, ts, t = _rast_inputs(g2)
    out = O.rasterize_to_pixels_2dgs(t["means2d"], t["ray_transforms"], t["colors"], t["opacities"], t["normals"], W, H,
                                     ts, t["offsets"], t["flatten_ids"], backgrounds=t["backgrounds"], distloss=True)
    rc, ra, rn, rd, rm, li, mi = out
    assert_close_ratio(rc, g2["rast_render_colors"], 1e-4, 5e-5, name="render_colors")
    assert_close_ratio(ra, g2["rast_render_alphas"], 1e-5, 2e-5, name="render_alphas")
    assert_close_ratio(rn, g2["rast_render_normals"], 1e-4, 5e-5, name="render_normals")
    g = O.rasterize_to_pixels_2dgs_bwd(
        t["means2d"], t["ray_transforms"], t["colors"], t["opacities"], t["normals"], W, H, ts, t["offsets"],
        t["flatten_ids"], rc, ra, li, mi, to_t(g2["rast_v_render_colors"]), to_t(g2["rast_v_render_alphas"]),
        to_t(g2["rast_v_render_normals"]), None, torch.zeros_like(ra), backgrounds=t["backgrounds"])
    for k in ("v_means2d", "v_ray_transforms", "v_colors", "v_opacities", "v_normals", "v_backgrounds"):
        assert_grad_close(g[k].reshape(g2["rast_" + k].shape), g2["rast_" + k], rel=5e-4, name=k)


def _torch_render_2dgs(m2, M, col, op, nrm, pairs, W, H, bg):
    """Differentiable fp64 restatement of RasterizeToPixels2DGSSerialBatchFwd.cu:356-428 on a fixed contributing set
    (`pairs[pixel]` = ordered surfel ids): colours, alpha, normals, distortion (L1 form, :409-421) and median depth
    (:423-428). One image."""
    rows_c, rows_a, rows_n, rows_d, rows_m = {}, {}, {}, {}, {}
    for (y, x), ids in pairs.items():
        ids = torch.tensor(ids)
        px, py = x + 0.5, y + 0.5
        Mi = M[ids]
        hu = px * Mi[:, 2] - Mi[:, 0]
        hv = py * Mi[:, 2] - Mi[:, 1]
        z = torch.linalg.cross(hu, hv, dim=-1)
        s = z[:, :2] / z[:, 2:3]
        gw3 = (s ** 2).sum(-1)
        dd = m2[ids] - torch.tensor([px, py], dtype=torch.float64)
        gw2 = 2.0 * (dd ** 2).sum(-1)
        sigma = 0.5 * torch.minimum(gw3, gw2)
        alpha = torch.clamp_max(op[ids] * torch.exp(-sigma), 0.99)
        T = torch.cumprod(torch.cat([torch.ones(1, dtype=torch.float64), 1 - alpha]), 0)
        w = alpha * T[:-1]
        depth = col[ids, -1]
        rows_c[(y, x)] = (w[:, None] * col[ids]).sum(0) + T[-1] * bg
        rows_a[(y, x)] = 1 - T[-1:]
        rows_n[(y, x)] = (w[:, None] * nrm[ids]).sum(0)
        wd = w * depth
        prev = torch.cumsum(wd, 0) - wd
        rows_d[(y, x)] = (2.0 * (wd * (1 - T[:-1]) - w * prev)).sum().reshape(1)
        k = int((T[:-1] > 0.5).nonzero().max())
        rows_m[(y, x)] = depth[k].reshape(1)
    # autograd-friendly assembly
    rc = torch.stack([torch.stack([rows_c.get((y, x), bg.double()) for x in range(W)]) for y in range(H)])
    ra = torch.stack([torch.stack([rows_a.get((y, x), torch.zeros(1, dtype=torch.float64)) for x in range(W)]) for y in range(H)])
    rn = torch.stack([torch.stack([rows_n.get((y, x), torch.zeros(3, dtype=torch.float64)) for x in range(W)]) for y in range(H)])
    rd = torch.stack([torch.stack([rows_d.get((y, x), torch.zeros(1, dtype=torch.float64)) for x in range(W)]) for y in range(H)])
    rm = torch.stack([torch.stack([rows_m.get((y, x), torch.zeros(1, dtype=torch.float64)) for x in range(W)]) for y in range(H)])
    return rc, ra, rn, rd, rm


def test_rasterize_2dgs_distortion_and_median_gradients_vs_autograd(O):
    """Distortion / median exist only in the reference's CUDA kernels: pin the oracle's restatement of them (forward
    AND backward) against torch autograd through an independent fp64 formulation."""
    sc, W, H = make_scene(N=60, C=1, width=24, height=16, seed=5, scale_range=(0.15, 0.5), z_range=(2.0, 5.0))
    rad, m2, d, M, nrm = O.fully_fused_projection_2dgs(sc["means"], sc["quats"], sc["scales"], sc["viewmats"], sc["Ks"],
                                                       W, H)
    op = sc["opacities"][None].contiguous()
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    _, ids, fl = O.isect_tiles(m2, rad, d, 16, tw, th, sort=True)
    off = O.isect_offset_encode(ids, 1, tw, th)
    g = torch.Generator().manual_seed(3)
    col = torch.cat([torch.rand(1, 60, 2, generator=g), d[..., None]], -1).contiguous()
    bg = torch.tensor([[0.3, 0.6, 0.0]])
    rc, ra, rn, rd, rm, li, mi = O.rasterize_to_pixels_2dgs(m2, M, col, op, nrm, W, H, 16, off, fl, backgrounds=bg,
                                                            distloss=True)
    gi, pi, ii = O.rasterize_to_indices_2dgs(m2, M, op, W, H, 16, off, fl)
    pairs = {}
    for gg, pp in zip(gi.tolist(), pi.tolist()):
        pairs.setdefault((pp // W, pp % W), []).append(gg)
    assert len(pairs) > 100
    leaves = [t[0].double().clone().requires_grad_(True) for t in (m2, M, col, op, nrm)]
    rc_t, ra_t, rn_t, rd_t, rm_t = _torch_render_2dgs(*leaves, pairs, W, H, bg[0].double())
    assert_close_ratio(rc[0], rc_t, 1e-4, 1e-5, name="render_colors")
    assert_close_ratio(ra[0], ra_t, 1e-4, 1e-5, name="render_alphas")
    assert_close_ratio(rn[0], rn_t, 1e-4, 1e-5, name="render_normals")
    assert_close_ratio(rd[0], rd_t, 1e-3, 1e-5, name="render_distort")
    assert_close_ratio(rm[0], rm_t, 1e-5, 1e-6, name="render_median")
    v = [torch.randn(t.shape, generator=g, dtype=torch.float64) for t in (rc_t, ra_t, rn_t, rd_t, rm_t)]
    loss = sum((a * b).sum() for a, b in zip((rc_t, ra_t, rn_t, rd_t, rm_t), v))
    ref = torch.autograd.grad(loss, leaves)
    got = O.rasterize_to_pixels_2dgs_bwd(m2, M, col, op, nrm, W, H, 16, off, fl, rc, ra, li, mi, v[0][None].float(),
                                         v[1][None].float(), v[2][None].float(), v[3][None].float(), v[4][None].float(),
                                         backgrounds=bg)
    for key, r in zip(("v_means2d", "v_ray_transforms", "v_colors", "v_opacities", "v_normals"), ref):
        assert_grad_close(got[key].reshape(r.shape), r, rel=1e-3, name=key)
    # densification signal = z-components of v_u_M / v_v_M times w_M.z (Bwd.cu:626-629)
    vrt = got["v_ray_transforms"].reshape(-1, 9)
    expect = np.stack([vrt[:, 2], vrt[:, 5]], -1) * M[0].reshape(-1, 9)[:, 8:9].double().numpy()
    assert_grad_close(got["v_densify"], expect, rel=1e-3, name="v_densify")
