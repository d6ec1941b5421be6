#!/usr/bin/env python
"""Golden vectors for the from-world rasterizer's HIT DISTANCE and NORMALS (and sample counts), from the reference's torch
statement (gsplat/cuda/_torch_impl_eval3d.py:264-495 `accumulate_eval3d(use_hit_distance=True, return_normals=True)`; per-sample
hit distance :173-223, facing normal :352-372) and its autograd. Writes tests/golden/eval3d_extras_ref.npz (inputs + the
REFERENCE's outputs and gradients); tests/test_gpu_eval3d.py replays them on gsx_raster_world_{fwd,bwd}_ex.
TEST INFRASTRUCTURE; run only where the reference checkout exists: python oracle/pin_eval3d_extras_against_reference.py
The same seeded scenes, tile lists and `nerfacc` restatement as oracle/pin_eval3d_against_reference.py."""
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
sys.path.insert(0, HERE)

import pin_eval3d_against_reference as base  # noqa: E402

CASES = {  # name: (N, C, W, H, tile, seed, backgrounds, channels, hit distance, normals)
    "hit": (150, 2, 40, 24, 16, 11, True, 4, True, False),
    "normals": (150, 1, 32, 32, 8, 12, False, 3, False, True),
    "both": (300, 2, 40, 24, 16, 13, True, 4, True, True),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "eval3d_extras_ref.npz"))
    args = ap.parse_args()
    sys.path.insert(0, args.ref)
    sys.dont_write_bytecode = True
    nerfacc = types.ModuleType("nerfacc")
    nerfacc.pack_info, nerfacc.render_weight_from_alpha, nerfacc.accumulate_along_rays = (
        base.pack_info, base.render_weight_from_alpha, base.accumulate_along_rays)
    sys.modules["nerfacc"] = nerfacc
    import gsplat_amd.csrc_shim as shim

    sys.modules["gsplat.csrc"] = shim
    from gsplat.cuda import _torch_impl_eval3d as R
    from oracle import eval3d as E
    from oracle import oracle as O

    gold = {}
    for name, (N, C, W, H, ts, seed, with_bg, D, hit, nrm) in CASES.items():
        means, quats, scales, opacities, colors3, viewmats, Ks = base.scene(N, C, W, H, seed)
        g = torch.Generator().manual_seed(100 + seed)
        colors = torch.cat([colors3, torch.rand(C, N, D - 3, generator=g)], -1) if D > 3 else colors3
        radii, means2d, depths, conics, _ = O.fully_fused_projection(means[None], None, quats[None], scales[None], viewmats[None],
                                                                     Ks[None], W, H)
        tw, th = math.ceil(W / ts), math.ceil(H / ts)
        _, isect_ids, flatten_ids = O.isect_tiles(means2d[0], radii[0] * 2, depths[0], ts, tw, th)
        offsets = O.isect_offset_encode(isect_ids, C, tw, th)
        rays = E.pinhole_rays(viewmats, Ks, W, H).clone().requires_grad_(True)  # the rays are a differentiable input too
        op_in = opacities[None].expand(C, N).contiguous()
        bg = torch.rand(C, D, generator=g) if with_bg else None
        rows, present = E.candidate_lists(offsets, flatten_ids, W, H, ts)
        P = C * H * W
        pix = torch.arange(P)[:, None].expand_as(rows)[present]
        row = rows[present]
        off = torch.cat([offsets.reshape(-1).long(), torch.tensor([flatten_ids.numel()])])
        ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        tile = (torch.arange(C)[:, None, None] * (th * tw) + ((ys // ts) * tw + xs // ts)[None]).reshape(-1)
        fidx = (off[tile][:, None] + torch.arange(rows.shape[1])[None, :])[present]
        leaves = [t.clone().requires_grad_(True) for t in (means, quats, scales, op_in, colors)]
        ren, alp, last, cnt, nor = R.accumulate_eval3d(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], row % N, pix % (H * W),
                                                       pix // (H * W), W, H, fidx, rays.reshape(C, H * W, 6),
                                                       use_hit_distance=hit, return_normals=nrm)
        if bg is not None:
            bg.requires_grad_(True)
            ren = ren + (1.0 - alp) * bg[:, None, None, :]
        v_r, v_a = torch.randn(ren.shape, generator=g), torch.randn(alp.shape, generator=g)
        loss = (ren * v_r).sum() + (alp * v_a).sum()
        v_n = None
        if nrm:
            v_n = torch.randn(nor.shape, generator=g)
            loss = loss + (nor * v_n).sum()
        loss.backward()
        print(f"case {name}: N={N} C={C} {W}x{H} tile {ts} D={D} hit={hit} normals={nrm}: mean samples/pixel "
              f"{float(cnt.float().mean()):.1f}, |render| max {float(ren.abs().max()):.3f}" + (f", |normals| max {float(nor.abs().max()):.3f}" if nrm else ""))
        for nm, a in zip(("means", "quats", "scales", "opacities", "colors"), leaves):
            gold[f"{name}.ref.v_{nm}"] = a.grad.numpy()
        gold[f"{name}.ref.v_rays"] = rays.grad.numpy()
        if bg is not None:
            gold[f"{name}.ref.v_backgrounds"] = bg.grad.numpy()
        rays, bg = rays.detach(), (None if bg is None else bg.detach())
        for k, v in dict(means=means, quats=quats, scales=scales, opacities=op_in, colors=colors, viewmats=viewmats, Ks=Ks, rays=rays,
                         isect_offsets=offsets, flatten_ids=flatten_ids, v_render=v_r, v_alpha=v_a).items():
            gold[f"{name}.{k}"] = v.numpy()
        if bg is not None:
            gold[f"{name}.backgrounds"] = bg.numpy()
        if nrm:
            gold[f"{name}.v_normals"], gold[f"{name}.ref.normals"] = v_n.numpy(), nor.detach().numpy()
        gold[f"{name}.ref.render"], gold[f"{name}.ref.alpha"] = ren.detach().numpy(), alp.detach().numpy()
        gold[f"{name}.ref.last_ids"], gold[f"{name}.ref.sample_counts"] = last.numpy(), cnt.numpy()
        gold[f"{name}.shape"] = np.array([N, C, W, H, ts, D, int(hit), int(nrm)])
    np.savez_compressed(args.out, **gold)
    print("wrote", args.out, os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()
