#!/usr/bin/env python
"""Pin oracle/eval3d.py (from-world compositing of 3DGUT) against the reference's torch statement
(gsplat/cuda/_torch_impl_eval3d.py:264-495 `accumulate_eval3d`, :91-132 `_generate_rays`) and write
tests/golden/eval3d_ref.npz (inputs + the REFERENCE's outputs and autograd gradients).

`accumulate_eval3d` composites explicit (gaussian, pixel, image) sample lists and needs the third-party `nerfacc`
(>= 0.5.3, setup.py:188, not vendored, not installable offline): its three entry points are restated here from their
published semantics - pack_info (start, count) per ray of a sorted index list; render_weight_from_alpha: w_i = alpha_i *
prod_{j<i in the ray} (1 - alpha_j) and that product; accumulate_along_rays: per-ray index_add. The sample lists are every
(pixel, Gaussian of the pixel's tile) pair in depth order, i.e. what rasterize_to_indices would enumerate before any test.
Run only where the reference checkout exists:   python oracle/pin_eval3d_against_reference.py [--ref /root/reference]
"""
import argparse
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


def pack_info(ray_indices, n_rays=None):
    n_rays = int(n_rays)
    cnt = torch.bincount(ray_indices, minlength=n_rays)
    start = torch.cumsum(cnt, 0) - cnt
    return torch.stack([start, cnt], dim=-1)


def render_weight_from_alpha(alphas, packed_info=None, ray_indices=None, n_rays=None, **_):
    if alphas.numel() == 0:
        return alphas, alphas
    if packed_info is not None:
        start, cnt = packed_info.unbind(-1)
        seg_start = torch.repeat_interleave(start, cnt)
    else:
        new = torch.ones_like(ray_indices, dtype=torch.bool)
        new[1:] = ray_indices[1:] != ray_indices[:-1]
        seg_start = torch.where(new)[0][torch.cumsum(new.long(), 0) - 1]
    logs = torch.log1p(-alphas.double())
    csum = torch.cumsum(logs, 0)
    excl = csum - logs
    trans = torch.exp(excl - excl[seg_start]).to(alphas.dtype)
    return alphas * trans, trans


def accumulate_along_rays(weights, values=None, ray_indices=None, n_rays=None):
    src = weights[:, None] if values is None else weights[:, None] * values
    return torch.zeros((n_rays, src.shape[-1]), dtype=src.dtype).index_add(0, ray_indices, src)


def scene(N, C, W, H, seed):
    g = torch.Generator().manual_seed(seed)
    f = 0.9 * W
    z = torch.rand(N, generator=g) * 5.0 + 1.0
    means = torch.stack([(torch.rand(N, generator=g) - 0.5) * 1.2 * W / f * z, (torch.rand(N, generator=g) - 0.5) * 1.2 * H / f * z, z], -1)
    quats = torch.nn.functional.normalize(torch.randn(N, 4, generator=g), dim=-1)
    scales = torch.exp(torch.randn(N, 3, generator=g) * 0.5 + math.log(0.12))
    opacities = torch.rand(N, generator=g) * 0.9 + 0.05
    colors = torch.rand(C, N, 3, generator=g)
    viewmats = torch.eye(4).repeat(C, 1, 1)
    for c in range(C):
        a = 0.06 * c
        viewmats[c, :3, :3] = torch.tensor([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]])
        viewmats[c, :3, 3] = torch.tensor([0.15 * c, -0.05 * c, 0.0])
    Ks = torch.tensor([[f, 0.0, W / 2], [0.0, f, H / 2], [0.0, 0.0, 1.0]]).repeat(C, 1, 1)
    return means, quats, scales, opacities, colors, viewmats, Ks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "eval3d_ref.npz"))
    args = ap.parse_args()
    sys.path.insert(0, args.ref)
    sys.dont_write_bytecode = True
    nerfacc = types.ModuleType("nerfacc")
    nerfacc.pack_info, nerfacc.render_weight_from_alpha, nerfacc.accumulate_along_rays = pack_info, render_weight_from_alpha, accumulate_along_rays
    sys.modules["nerfacc"] = nerfacc
    import gsplat_amd.csrc_shim as shim

    sys.modules["gsplat.csrc"] = shim
    from gsplat.cuda import _torch_impl_eval3d as R
    from oracle import eval3d as E
    from oracle import oracle as O

    gold = {}
    for name, (N, C, W, H, ts, seed, with_bg) in {"a": (150, 2, 40, 24, 16, 1, True), "b": (400, 1, 32, 32, 8, 2, False)}.items():
        means, quats, scales, opacities, colors, viewmats, Ks = scene(N, C, W, H, seed)
        # tile lists from the classic projection + intersection oracle (any superset of the contributing Gaussians works)
        radii, means2d, depths, conics, _ = O.fully_fused_projection(means[None], None, quats[None], scales[None], viewmats[None],
                                                                     Ks[None], W, H)
        tw, th = math.ceil(W / ts), math.ceil(H / ts)
        radii = radii[0] * 2  # generous boxes: the from-world response is not the 2D conic's
        _, isect_ids, flatten_ids = O.isect_tiles(means2d[0], radii, depths[0], ts, tw, th)
        offsets = O.isect_offset_encode(isect_ids, C, tw, th)
        rays = E.pinhole_rays(viewmats, Ks, W, H)
        # the reference's own ray generator agrees
        from gsplat.cuda._torch_cameras import _BaseCameraModel
        from gsplat.cuda._wrapper import RollingShutterType
        cam = _BaseCameraModel.create(width=W, height=H, camera_model="pinhole", principal_points=Ks[:, :2, 2],
                                      focal_lengths=torch.stack([Ks[:, 0, 0], Ks[:, 1, 1]], -1), rs_type=RollingShutterType.GLOBAL)
        rays_ref = R._generate_rays(cam, W, H, viewmats)
        rays_ref = rays_ref.reshape(C, H, W, 6)
        assert float((rays_ref - rays).abs().max()) < 2e-5, float((rays_ref - rays).abs().max())

        op_in = opacities[None].expand(C, N).contiguous()
        bg = torch.rand(C, 3, generator=torch.Generator().manual_seed(9)) if with_bg else None
        rows, present = E.candidate_lists(offsets, flatten_ids, W, H, ts)
        P = C * H * W
        pix = torch.arange(P)[:, None].expand_as(rows)[present]
        row = rows[present]
        off = torch.cat([offsets.reshape(-1).long(), torch.tensor([flatten_ids.numel()])])
        ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        tile = (torch.arange(C)[:, None, None] * (th * tw) + ((ys // ts) * tw + xs // ts)[None]).reshape(-1)
        fidx = (off[tile][:, None] + torch.arange(rows.shape[1])[None, :])[present]

        leaves_r = [t.clone().requires_grad_(True) for t in (means, quats, scales, op_in, colors)]
        ren, alp, last, cnt, _ = R.accumulate_eval3d(leaves_r[0], leaves_r[1], leaves_r[2], leaves_r[3], leaves_r[4], row % N,
                                                    pix % (H * W), pix // (H * W), W, H, fidx, rays.reshape(C, H * W, 6))
        if bg is not None:
            ren = ren + (1.0 - alp) * bg[:, None, None, :]
        g = torch.Generator().manual_seed(5)
        v_r, v_a = torch.randn(ren.shape, generator=g), torch.randn(alp.shape, generator=g)
        ((ren * v_r).sum() + (alp * v_a).sum()).backward()

        leaves_o = [t.clone().requires_grad_(True) for t in (means, quats, scales, op_in, colors)]
        ren_o, alp_o, last_o = E.rasterize_to_pixels_eval3d(leaves_o[0], leaves_o[1], leaves_o[2], leaves_o[4], leaves_o[3], rays, W, H,
                                                            ts, offsets, flatten_ids, backgrounds=bg)
        ((ren_o * v_r).sum() + (alp_o * v_a).sum()).backward()
        d_ren, d_alp = float((ren - ren_o).abs().max()), float((alp - alp_o).abs().max())
        same_last = float((last == last_o).float().mean())
        print(f"case {name}: N={N} C={C} {W}x{H} tile {ts}: pairs {int(present.sum())}, max|d render| {d_ren:.2e}, max|d alpha| "
              f"{d_alp:.2e}, last_ids equal {same_last:.4f}, mean samples/pixel {float(cnt.float().mean()):.1f}")
        assert d_ren < 2e-5 and d_alp < 2e-5 and same_last > 0.999
        for nm, a, b in zip(("means", "quats", "scales", "opacities", "colors"), leaves_r, leaves_o):
            scale_ = float(a.grad.abs().max()) + 1e-12
            err = float((a.grad - b.grad).abs().max()) / scale_
            print(f"    grad {nm:9s} max err / max |grad| = {err:.2e}")
            assert err < 2e-4, nm
            gold[f"{name}.ref.v_{nm}"] = a.grad.numpy()
        for k, v in dict(means=means, quats=quats, scales=scales, opacities=op_in, colors=colors, viewmats=viewmats, Ks=Ks, rays=rays,
                         isect_offsets=offsets, flatten_ids=flatten_ids, v_render=v_r, v_alpha=v_a).items():
            gold[f"{name}.{k}"] = v.numpy()
        if bg is not None:
            gold[f"{name}.backgrounds"] = bg.numpy()
        gold[f"{name}.ref.render"], gold[f"{name}.ref.alpha"], gold[f"{name}.ref.last_ids"] = ren.detach().numpy(), alp.detach().numpy(), last.numpy()
        gold[f"{name}.shape"] = np.array([N, C, W, H, ts])
    np.savez_compressed(args.out, **gold)
    print("wrote", args.out, os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()
