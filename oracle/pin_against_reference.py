#!/usr/bin/env python
"""Pin the CPU oracle against the reference's own Python implementation and write golden vectors.

Run ONLY where the reference checkout exists (this container: /root/reference). It

1. imports ``gsplat`` from the reference (CPU, ``_C = None``) and checks every oracle function
   against the reference function it restates (``gsplat/cuda/_torch_impl.py``, ``_math.py``);
2. for the compositing stage, drives the reference's ``accumulate()`` (``_torch_impl.py:713-811``)
   with the oracle's contributing-pair list and a restated ``nerfacc`` (the third-party package
   ``nerfacc>=0.5.3`` named in the reference's setup.py:188 is not in this image and not under
   /root/reference; its two published functions are restated below), and takes gradients with
   torch autograd through that reference code;
3. stores small input/expected-output fixtures in ``tests/golden/*.npz`` so that the checks can be
   repeated on machines without the reference (``tests/test_oracle_golden.py``, and the GPU parity
   tests which compare the HIP kernels with the same expected outputs).

Usage:  python oracle/pin_against_reference.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import oracle as O  # noqa: E402


# ----------------------------------------------------------------------------------------------
# restated nerfacc (published semantics): per ray, w_i = alpha_i * prod_{j<i}(1 - alpha_j)
# ----------------------------------------------------------------------------------------------
def _segment_starts(ray_indices):
    n = ray_indices.shape[0]
    new = torch.ones(n, dtype=torch.bool)
    if n > 1:
        new[1:] = ray_indices[1:] != ray_indices[:-1]
    seg_id = torch.cumsum(new.long(), 0) - 1
    start = torch.where(new)[0]
    return seg_id, start


def render_weight_from_alpha(alphas, ray_indices=None, n_rays=None, **_):
    if alphas.numel() == 0:
        return alphas, alphas
    seg_id, start = _segment_starts(ray_indices)
    logs = torch.log1p(-alphas.double())
    csum = torch.cumsum(logs, 0)
    excl = csum - logs  # exclusive cumsum
    excl = excl - excl[start][seg_id]
    trans = torch.exp(excl).to(alphas.dtype)
    return alphas * trans, trans


def accumulate_along_rays(weights, values=None, ray_indices=None, n_rays=None):
    if values is None:
        src = weights[:, None]
    else:
        src = weights[:, None] * values
    out = torch.zeros((n_rays, src.shape[-1]), dtype=src.dtype)
    return out.index_add(0, ray_indices, src)


def install_nerfacc_stub():
    m = types.ModuleType("nerfacc")
    m.render_weight_from_alpha = render_weight_from_alpha
    m.accumulate_along_rays = accumulate_along_rays
    sys.modules["nerfacc"] = m


def close(name, a, b, rtol, atol):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = (err > tol).sum().item()
    print(f"  {name:34s} max|err|={err.max().item() if err.numel() else 0:.3e}  bad={bad}/{err.numel()}")
    assert bad == 0, f"{name}: {bad} elements out of tolerance"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    args = ap.parse_args()
    sys.path.insert(0, args.ref)
    sys.dont_write_bytecode = True
    install_nerfacc_stub()
    import gsplat  # noqa: F401  (reference, CPU only)
    from gsplat.cuda import _torch_impl as R
    from gsplat.cuda._math import _quat_scale_to_covar_preci

    os.makedirs(args.out, exist_ok=True)
    torch.manual_seed(42)
    gold = {}

    # ---- scene: reference test asset, cropped like gsplat/_helper.py:51-102, subsampled ----
    data = np.load(os.path.join(args.ref, "assets", "test_garden.npz"))
    means_all = torch.from_numpy(data["means3d"]).float()
    colors_all = torch.from_numpy(data["colors"] / 255.0).float()
    sel = ((means_all >= -2) & (means_all <= 2)).all(-1)
    means_all, colors_all = means_all[sel], colors_all[sel]
    idx = torch.randperm(means_all.shape[0])[:3000]
    means = means_all[idx].contiguous()
    colors = colors_all[idx].contiguous()
    N = means.shape[0]
    viewmats = torch.from_numpy(data["viewmats"]).float()[:2].contiguous()
    scale = 0.25  # quarter resolution keeps the fixtures small
    width, height = int(data["width"].item() * scale), int(data["height"].item() * scale)
    Ks = torch.from_numpy(data["Ks"]).float()[:2].clone()
    Ks[:, :2, :] *= scale
    C = viewmats.shape[0]
    scales = torch.rand((N, 3)) * (0.02 - 1e-4) + 1e-4
    # enlarge a bit so that quarter-res splats still cover pixels
    scales = scales * 3.0
    quats = F.normalize(torch.randn((N, 4)), dim=-1)
    opacities = torch.rand((N,))

    # ---- 1. quat/scale -> covar/preci ---------------------------------------------------
    print("[1] quat_scale_to_covar_preci vs gsplat/cuda/_math.py:689")
    for triu in (False, True):
        q = quats.clone().requires_grad_(True)
        s = scales.clone().requires_grad_(True)
        c_o, p_o = O.quat_scale_to_covar_preci(q, s, True, True, triu)
        c_r, p_r = _quat_scale_to_covar_preci(q, s, True, True, triu)
        close(f"covars triu={triu}", c_o, c_r, 1e-5, 1e-7)
        close(f"precis triu={triu}", p_o, p_r, 1e-4, 1e-2)
    gold["qs_quats"], gold["qs_scales"] = quats[:256].numpy(), scales[:256].numpy()
    c_r, p_r = _quat_scale_to_covar_preci(quats[:256], scales[:256], True, True, True)
    gold["qs_covars_triu"], gold["qs_precis_triu"] = c_r.numpy(), p_r.numpy()

    # ---- 2. projection ------------------------------------------------------------------
    print("[2] fully_fused_projection vs _torch_impl.py:262-352")
    covars_full = _quat_scale_to_covar_preci(quats, scales, True, False, False)[0]
    for cam in ("pinhole", "ortho", "fisheye"):
        mr = means.clone().requires_grad_(True)
        qr = quats.clone().requires_grad_(True)
        sr = scales.clone().requires_grad_(True)
        vr = viewmats.clone().requires_grad_(True)
        cov_r = _quat_scale_to_covar_preci(qr, sr, True, False, False)[0]
        rad_r, m2_r, d_r, con_r, comp_r = R._fully_fused_projection(
            mr, cov_r, vr, Ks, width, height, eps2d=0.3, near_plane=0.01, far_plane=1e10,
            calc_compensations=True, camera_model=cam)
        mo = means.clone().requires_grad_(True)
        qo = quats.clone().requires_grad_(True)
        so = scales.clone().requires_grad_(True)
        vo = viewmats.clone().requires_grad_(True)
        rad_o, m2_o, d_o, con_o, comp_o = O.fully_fused_projection(
            mo[None], None, qo[None], so[None], vo[None], Ks[None], width, height, 0.3, 0.01, 1e10, 0.0, True, cam,
            None)
        rad_o, m2_o, d_o, con_o, comp_o = rad_o[0], m2_o[0], d_o[0], con_o[0], comp_o[0]
        valid = (rad_r > 0).all(-1) & (rad_o > 0).all(-1)
        agree = ((rad_r > 0).all(-1) == (rad_o > 0).all(-1)).float().mean().item()
        print(f"  [{cam}] visibility agreement {agree:.5f}, valid {valid.sum().item()}/{valid.numel()}")
        assert agree > 0.999
        assert (rad_r[valid] - rad_o[valid]).abs().max().item() <= 1
        close(f"{cam} means2d", m2_o[valid], m2_r[valid], 1e-4, 1e-4)
        close(f"{cam} depths", d_o[valid], d_r[valid], 1e-4, 1e-4)
        close(f"{cam} conics", con_o[valid], con_r[valid], 1e-4, 1e-4)
        close(f"{cam} compensations", comp_o[valid], comp_r[valid], 1e-4, 1e-3)
        # gradients through both
        w_m2 = torch.randn_like(m2_r)
        w_d = torch.randn_like(d_r)
        w_c = torch.randn_like(con_r) * 1e-2
        w_k = torch.randn_like(comp_r)
        vm = valid[..., None].float()
        loss_r = (m2_r * w_m2 * vm).sum() + (d_r * w_d * valid).sum() + (con_r * w_c * vm).sum() + (comp_r * w_k * valid).sum()
        loss_o = (m2_o * w_m2 * vm).sum() + (d_o * w_d * valid).sum() + (con_o * w_c * vm).sum() + (comp_o * w_k * valid).sum()
        g_r = torch.autograd.grad(loss_r, [mr, qr, sr, vr])
        g_o = torch.autograd.grad(loss_o, [mo, qo, so, vo])
        for nm, a, b in zip(("v_means", "v_quats", "v_scales", "v_viewmats"), g_o, g_r):
            sc = b.abs().max().item() + 1e-12
            close(f"{cam} {nm} (rel to max)", a / sc, b / sc, 0.0, 2e-3)
        if cam == "pinhole":
            gold.update(proj_means=means.numpy(), proj_quats=quats.numpy(), proj_scales=scales.numpy(),
                        proj_viewmats=viewmats.numpy(), proj_Ks=Ks.numpy(), proj_wh=np.array([width, height]),
                        proj_opacities=opacities.numpy())
        gold[f"proj_{cam}_radii"] = rad_r.numpy().astype(np.int32)
        gold[f"proj_{cam}_means2d"] = m2_r.detach().numpy()
        gold[f"proj_{cam}_depths"] = d_r.detach().numpy()
        gold[f"proj_{cam}_conics"] = con_r.detach().numpy()
        gold[f"proj_{cam}_comps"] = comp_r.detach().numpy()
        gold[f"proj_{cam}_w"] = np.concatenate([w_m2.numpy().reshape(C, N, -1), w_d.numpy()[..., None],
                                                w_c.numpy().reshape(C, N, -1), w_k.numpy()[..., None]], -1)
        gold[f"proj_{cam}_valid"] = valid.numpy()
        for nm, b in zip(("v_means", "v_quats", "v_scales", "v_viewmats"), g_r):
            gold[f"proj_{cam}_{nm}"] = b.numpy()

    # ---- 2b. proj() (projection_ewa_simple) vs _torch_impl._persp_proj / _ortho_proj / _fisheye_proj ----------
    print("[2b] proj vs _torch_impl.py:53-260")
    mc, cc = R._world_to_cam(means, covars_full, viewmats)
    front = mc[..., 2] > 0.2
    for cam, fn in (("pinhole", R._persp_proj), ("ortho", R._ortho_proj), ("fisheye", R._fisheye_proj)):
        m_r, c_r = fn(mc, cc, Ks, width, height)
        m_o, c_o = O.proj(mc, cc, Ks, width, height, cam)
        close(f"proj {cam} means2d", m_o[front], m_r[front], 1e-4, 1e-3)
        close(f"proj {cam} covars2d (rel)", c_o[front] / (c_r[front].abs().max() + 1e-12),
              c_r[front] / (c_r[front].abs().max() + 1e-12), 0.0, 1e-5)

    # ---- 3. spherical harmonics ---------------------------------------------------------
    print("[3] spherical_harmonics vs _torch_impl.py:1052-1067")
    coeffs = torch.randn(N, 25, 3) * 0.3
    campos = -torch.einsum("cji,cj->ci", viewmats[:, :3, :3], viewmats[:, :3, 3])
    dirs = means[None] - campos[:, None]
    for deg in range(5):
        c_r = R._spherical_harmonics(deg, dirs, coeffs)
        c_o = O.spherical_harmonics(deg, means[None], viewmats[None], coeffs)[0]
        close(f"sh deg {deg}", c_o, c_r, 1e-5, 1e-5)
        gold[f"sh_colors_deg{deg}"] = c_r.numpy()
    gold["sh_coeffs"] = coeffs.numpy()

    # ---- 4. tile intersection (AABB mode is what the reference's torch code restates) ----
    print("[4] isect_tiles / isect_offset_encode vs _torch_impl.py:356-481 (exact)")
    rad_o, m2_o, d_o, con_o, comp_o = O.fully_fused_projection(
        means[None], None, quats[None], scales[None], viewmats[None], Ks[None], width, height, 0.3, 0.01, 1e10, 0.0,
        False, "pinhole", opacities[None])
    rad, m2, dep, con = rad_o[0], m2_o[0].detach(), d_o[0].detach(), con_o[0].detach()
    tile_size = 16
    tw, th = math.ceil(width / tile_size), math.ceil(height / tile_size)
    sub = slice(0, 1200)  # the reference loop is pure Python
    t_r, ids_r, fl_r = R._isect_tiles(m2[:, sub], rad[:, sub].float(), dep[:, sub], tile_size, tw, th, sort=True)
    t_o, ids_o, fl_o = O.isect_tiles(m2[:, sub], rad[:, sub], dep[:, sub], tile_size, tw, th, sort=True)
    assert torch.equal(t_r.int(), t_o.int()), "tiles_per_gauss differ"
    assert torch.equal(ids_r, ids_o), "isect_ids differ"
    assert torch.equal(fl_r.int(), fl_o.int()), "flatten_ids differ"
    off_r = R._isect_offset_encode(ids_r, C, tw, th)
    off_o = O.isect_offset_encode(ids_o, C, tw, th)
    assert torch.equal(off_r.int(), off_o.int()), "offsets differ"
    print(f"  exact match: {ids_o.numel()} intersections, {C * tw * th} tiles")
    gold.update(isect_means2d=m2[:, sub].numpy(), isect_radii=rad[:, sub].numpy().astype(np.int32),
                isect_depths=dep[:, sub].numpy(), isect_conics=con[:, sub].numpy(),
                isect_opacities=opacities[None, sub].expand(C, -1).numpy().copy(),
                isect_tile=np.array([tile_size, tw, th]), isect_tiles_per_gauss=t_r.numpy().astype(np.int32),
                isect_ids=ids_r.numpy(), isect_flatten_ids=fl_r.numpy().astype(np.int32),
                isect_offsets=off_r.numpy().astype(np.int32))
    # ellipse (AccuTile) mode has no Python restatement in the reference; record the oracle's own
    # output so HIP-vs-oracle exactness is also checked against a committed vector. It must be a
    # subset of the AABB result (it only removes tiles) and must keep every tile the forward touches.
    op_c = opacities[None].expand(C, -1)
    t_a, ids_a, fl_a = O.isect_tiles(m2[:, sub], rad[:, sub], dep[:, sub], tile_size, tw, th, sort=True,
                                     conics=con[:, sub], opacities=op_c[:, sub])
    assert (t_a <= t_o).all(), "ellipse mode must not add tiles"
    gold.update(isect_accu_tiles_per_gauss=t_a.numpy(), isect_accu_ids=ids_a.numpy(), isect_accu_flatten_ids=fl_a.numpy())
    print(f"  ellipse mode keeps {ids_a.numel()}/{ids_o.numel()} intersections")

    # ---- 5. compositing forward/backward vs reference accumulate() ------------------------
    print("[5] rasterize_to_pixels vs accumulate() (_torch_impl.py:713-811) + autograd")
    for mode in ("aabb", "accu"):
        if mode == "aabb":
            tpg, ids, fl = O.isect_tiles(m2, rad, dep, tile_size, tw, th, sort=True)
        else:
            tpg, ids, fl = O.isect_tiles(m2, rad, dep, tile_size, tw, th, sort=True, conics=con, opacities=op_c)
        off = O.isect_offset_encode(ids, C, tw, th)
        cols = colors[None].expand(C, -1, -1).contiguous()
        bg = torch.rand(C, 3)
        rc_o, ra_o, li_o = O.rasterize_to_pixels(m2, con, cols, op_c, width, height, tile_size, off, fl, backgrounds=bg)
        g_ids, p_ids, i_ids = O.rasterize_to_indices(m2, con, op_c, width, height, tile_size, off, fl)
        m2g = m2.clone().requires_grad_(True)
        cong = con.clone().requires_grad_(True)
        colg = cols.clone().requires_grad_(True)
        opg = op_c.clone().contiguous().requires_grad_(True)
        bgg = bg.clone().requires_grad_(True)
        rc_r, ra_r = R.accumulate(m2g, cong, opg, colg, g_ids, p_ids, i_ids, width, height)
        rc_r = rc_r + bgg[:, None, None, :] * (1.0 - ra_r)
        close(f"[{mode}] render_colors", rc_o, rc_r, 1e-5, 2e-5)
        close(f"[{mode}] render_alphas", ra_o, ra_r, 1e-5, 2e-5)
        v_rc = torch.randn_like(rc_r)
        v_ra = torch.randn_like(ra_r)
        loss = (rc_r * v_rc).sum() + (ra_r * v_ra).sum()
        g = torch.autograd.grad(loss, [m2g, cong, colg, opg, bgg])
        gr = O.rasterize_to_pixels_bwd(m2, con, cols, op_c, width, height, tile_size, off, fl, ra_o, li_o, v_rc, v_ra,
                                       backgrounds=bg)
        for nm, key, b in zip(("v_means2d", "v_conics", "v_colors", "v_opacities", "v_backgrounds"),
                              ("v_means2d", "v_conics", "v_colors", "v_opacities", "v_backgrounds"), g):
            a = torch.from_numpy(gr[key]).reshape(b.shape)
            sc = b.abs().max().item() + 1e-12
            close(f"[{mode}] {nm} (rel to max)", a / sc, b / sc, 0.0, 5e-4)
        if mode == "accu":
            gold.update(rast_means2d=m2.numpy(), rast_conics=con.numpy(), rast_colors=cols.numpy(),
                        rast_opacities=op_c.numpy().copy(), rast_backgrounds=bg.numpy(),
                        rast_wh=np.array([width, height, tile_size]), rast_offsets=off.numpy(),
                        rast_flatten_ids=fl.numpy(), rast_render_colors=rc_r.detach().numpy(),
                        rast_render_alphas=ra_r.detach().numpy(), rast_last_ids=li_o.numpy(),
                        rast_v_render_colors=v_rc.numpy(), rast_v_render_alphas=v_ra.numpy(),
                        rast_v_means2d=g[0].numpy(), rast_v_conics=g[1].numpy(), rast_v_colors=g[2].numpy(),
                        rast_v_opacities=g[3].numpy(), rast_v_backgrounds=g[4].numpy())

    # ---- 6. 2DGS: projection vs _torch_impl_2dgs.py:27-108, compositing vs accumulate_2dgs (:111-212) -------
    print("[6] 2DGS projection / rasterize_to_pixels_2dgs vs gsplat/cuda/_torch_impl_2dgs.py + autograd")
    from gsplat.cuda import _torch_impl_2dgs as R2

    gold2 = {}
    sc2 = scales.clone()
    sc2[:, 2] = 1.0  # the CUDA kernel ignores the third scale (Projection2DGSFused.cu:167); the torch ref does not
    mr, qr, sr, vr = (t.clone().requires_grad_(True) for t in (means, quats, sc2, viewmats))
    rad_r, m2_r, d_r, M_r, n_r = R2._fully_fused_projection_2dgs(mr, qr, sr, vr, Ks, width, height, 0.01, 1e10)
    mo, qo, so, vo = (t.clone().requires_grad_(True) for t in (means, quats, scales, viewmats))
    rad_o, m2_o, d_o, M_o, n_o = O.fully_fused_projection_2dgs(mo, qo, so, vo, Ks, width, height, 0.01, 1e10)
    valid = (rad_r > 0).all(-1) & (rad_o > 0).all(-1)
    agree = ((rad_r > 0).all(-1) == (rad_o > 0).all(-1)).float().mean().item()
    print(f"  visibility agreement {agree:.5f}, valid {valid.sum().item()}/{valid.numel()}")
    assert agree > 0.999
    assert (rad_r[valid] - rad_o[valid]).abs().max().item() <= 1
    close("2dgs means2d", m2_o[valid], m2_r[valid], 1e-4, 1e-3)
    close("2dgs depths", d_o[valid], d_r[valid], 1e-5, 1e-5)
    close("2dgs ray_transforms", M_o[valid], M_r[valid], 1e-4, 1e-3)
    close("2dgs normals", n_o[valid], n_r[valid], 1e-5, 1e-5)
    w_m2, w_d, w_M, w_n = torch.randn_like(m2_r), torch.randn_like(d_r), torch.randn_like(M_r), torch.randn_like(n_r)
    vm = valid.float()

    def loss2(m2_, d_, M_, n_):
        return ((m2_ * w_m2).sum(-1) * vm).sum() + (d_ * w_d * vm).sum() + ((M_ * w_M).sum((-1, -2)) * vm).sum() \
            + ((n_ * w_n).sum(-1) * vm).sum()

    g_r = torch.autograd.grad(loss2(m2_r, d_r, M_r, n_r), [mr, qr, sr, vr])
    g_o = torch.autograd.grad(loss2(m2_o, d_o, M_o, n_o), [mo, qo, so, vo])
    for nm, a, b in zip(("v_means", "v_quats", "v_scales", "v_viewmats"), g_o, g_r):
        if nm == "v_scales":
            a, b = a[:, :2], b[:, :2]  # third scale: ignored by the kernel
        scl = b.abs().max().item() + 1e-12
        close(f"2dgs {nm} (rel to max)", a / scl, b / scl, 0.0, 2e-3)
    gold2.update(means=means.numpy(), quats=quats.numpy(), scales=scales.numpy(), viewmats=viewmats.numpy(),
                 Ks=Ks.numpy(), wh=np.array([width, height]), proj_radii=rad_r.numpy().astype(np.int32),
                 proj_means2d=m2_r.detach().numpy(), proj_depths=d_r.detach().numpy(),
                 proj_ray_transforms=M_r.detach().numpy(), proj_normals=n_r.detach().numpy(),
                 proj_valid=valid.numpy(), proj_w_means2d=w_m2.numpy(), proj_w_depths=w_d.numpy(),
                 proj_w_ray_transforms=w_M.numpy(), proj_w_normals=w_n.numpy(), proj_v_means=g_r[0].numpy(),
                 proj_v_quats=g_r[1].numpy(), proj_v_scales=g_r[2].numpy(), proj_v_viewmats=g_r[3].numpy())

    rad2, m22, dep2, M2, nrm2 = rad_o, m2_o.detach(), d_o.detach(), M_o.detach(), n_o.detach()
    tpg, ids, fl = O.isect_tiles(m22, rad2, dep2, tile_size, tw, th, sort=True)  # 2DGS always uses the AABB test
    off = O.isect_offset_encode(ids, C, tw, th)
    cols = torch.cat([colors[None].expand(C, -1, -1), dep2[..., None]], -1).contiguous()  # RGB + depth channel
    bg = torch.cat([torch.rand(C, 3), torch.zeros(C, 1)], -1)
    out_o = O.rasterize_to_pixels_2dgs(m22, M2, cols, op_c, nrm2, width, height, tile_size, off, fl, backgrounds=bg,
                                       distloss=True)
    rc_o, ra_o, rn_o, rd_o, rm_o, li_o, mi_o = out_o
    g_ids, p_ids, i_ids = O.rasterize_to_indices_2dgs(m22, M2, op_c, width, height, tile_size, off, fl)
    m2g, Mg, colg, nrg = (t.clone().requires_grad_(True) for t in (m22, M2, cols, nrm2))
    opg = op_c.clone().contiguous().requires_grad_(True)
    bgg = bg.clone().requires_grad_(True)
    rc_r, ra_r, rn_r = R2.accumulate_2dgs(m2g, Mg, opg, colg, nrg, g_ids, p_ids, i_ids, width, height)
    rc_r = rc_r + bgg[:, None, None, :] * (1.0 - ra_r)
    close("2dgs render_colors", rc_o, rc_r, 1e-4, 5e-5)
    close("2dgs render_alphas", ra_o, ra_r, 1e-5, 2e-5)
    close("2dgs render_normals", rn_o, rn_r, 1e-4, 5e-5)
    v_rc, v_ra, v_rn = torch.randn_like(rc_r), torch.randn_like(ra_r), torch.randn_like(rn_r)
    g = torch.autograd.grad((rc_r * v_rc).sum() + (ra_r * v_ra).sum() + (rn_r * v_rn).sum(),
                            [m2g, Mg, colg, opg, nrg, bgg])
    gr = O.rasterize_to_pixels_2dgs_bwd(m22, M2, cols, op_c, nrm2, width, height, tile_size, off, fl, rc_o, ra_o, li_o,
                                        mi_o, v_rc, v_ra, v_rn, None, torch.zeros_like(ra_o), backgrounds=bg)
    for key, b in zip(("v_means2d", "v_ray_transforms", "v_colors", "v_opacities", "v_normals", "v_backgrounds"), g):
        a = torch.from_numpy(gr[key]).reshape(b.shape)
        scl = b.abs().max().item() + 1e-12
        close(f"2dgs {key} (rel to max)", a / scl, b / scl, 0.0, 5e-4)
    gold2.update(rast_means2d=m22.numpy(), rast_ray_transforms=M2.numpy(), rast_colors=cols.numpy(),
                 rast_opacities=op_c.numpy().copy(), rast_normals=nrm2.numpy(), rast_backgrounds=bg.numpy(),
                 rast_wh=np.array([width, height, tile_size]), rast_offsets=off.numpy(), rast_flatten_ids=fl.numpy(),
                 rast_render_colors=rc_r.detach().numpy(), rast_render_alphas=ra_r.detach().numpy(),
                 rast_render_normals=rn_r.detach().numpy(), rast_v_render_colors=v_rc.numpy(),
                 rast_v_render_alphas=v_ra.numpy(), rast_v_render_normals=v_rn.numpy(), rast_v_means2d=g[0].numpy(),
                 rast_v_ray_transforms=g[1].numpy(), rast_v_colors=g[2].numpy(), rast_v_opacities=g[3].numpy(),
                 rast_v_normals=g[4].numpy(), rast_v_backgrounds=g[5].numpy())
    path2 = os.path.join(args.out, "garden_quarter_2dgs.npz")
    np.savez_compressed(path2, **gold2)
    print(f"wrote {path2} ({os.path.getsize(path2) / 1e6:.2f} MB)")

    path = os.path.join(args.out, "garden_quarter.npz")
    np.savez_compressed(path, **gold)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)")
    print("ORACLE PINNED: all checks passed")


if __name__ == "__main__":
    main()
