#!/usr/bin/env python
"""Pin the sparse pixel-set restatements of the oracle (sparse_tile_layout, isect_tiles_sparse) against the reference's
own looped Python references (gsplat/cuda/_torch_impl.py:485-604 `_isect_tiles_sparse`, :612-708
`_build_sparse_tile_layout`) and write tests/golden/sparse_ref.npz (inputs + the REFERENCE's outputs).

Scenes follow the reference's sparse tests (tests/sparse_test_helpers.py:37-58 make_scene: isotropic 2D Gaussians,
radii = ceil(3 sigma)) with CPU generators. Run only where the reference checkout exists:
    python oracle/pin_sparse_against_reference.py [--ref /root/reference]
"""
import argparse
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from pin_against_reference import install_nerfacc_stub  # noqa: E402
from oracle import oracle as O  # noqa: E402


def scene(C, N, W, H, seed):
    g = torch.Generator().manual_seed(seed)
    u = lambda *s: torch.rand(*s, generator=g)  # noqa: E731
    means2d = torch.stack([u(C, N) * W, u(C, N) * H], -1)
    r = u(C, N) * 6.0 + 2.0
    rb = (r * 3.0).ceil().to(torch.int32)
    radii = torch.stack([rb, rb], -1).contiguous()
    radii[:, ::17] = 0  # some culled rows
    depths = u(C, N) * 9.9 + 0.1
    return means2d.contiguous(), radii, depths.contiguous(), g


def pixel_set(P, I, W, H, g):
    flat = torch.randperm(I * H * W, generator=g)[:P]
    img, rem = flat // (H * W), flat % (H * W)
    return torch.stack([rem // W, rem % W], -1).to(torch.int32), img.to(torch.int32)


def u64(t):
    return t.view(torch.int64).numpy().view(np.uint64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden", "sparse_ref.npz"))
    args = ap.parse_args()
    sys.path.insert(0, args.ref)
    sys.dont_write_bytecode = True
    install_nerfacc_stub()
    from gsplat.cuda import _torch_impl as R

    gold = {}
    cases = [("a", 2, 400, 150, 100, 16, 700), ("b", 1, 300, 70, 50, 8, 300), ("c", 3, 200, 45, 37, 4, 150),
             ("d", 1, 100, 64, 64, 16, 64 * 64)]
    for name, C, N, W, H, ts, P in cases:
        tw, th = math.ceil(W / ts), math.ceil(H / ts)
        means2d, radii, depths, g = scene(C, N, W, H, seed=len(name) + N)
        pixels, image_ids = pixel_set(P, C, W, H, g)
        lay_r = R._build_sparse_tile_layout(pixels, image_ids, C, ts, tw, th)
        lay_o = O.sparse_tile_layout(pixels, image_ids, C, ts, tw, th)
        for k, (r, o) in enumerate(zip(lay_r, lay_o)):
            r = u64(r) if r.dtype == torch.uint64 else r.numpy()
            assert np.array_equal(r, o), (name, "layout output", k)
        act, tmask = lay_r[0], lay_r[1]
        off_r, fl_r = R._isect_tiles_sparse(means2d, radii, depths, tmask, act, C, ts, tw, th)
        off_o, fl_o = O.isect_tiles_sparse(means2d, radii, depths, tmask, act, C, ts, tw, th)
        assert torch.equal(off_r, off_o) and torch.equal(fl_r, fl_o), (name, "dense isect")
        # packed rows: visible rows only, image ids int32 as the reference allows
        vis = (radii > 0).all(-1)
        ci, _ = torch.where(vis)
        offp_r, flp_r = R._isect_tiles_sparse(means2d[vis], radii[vis], depths[vis], tmask, act, C, ts, tw, th,
                                              image_ids=ci.to(torch.int32))
        offp_o, flp_o = O.isect_tiles_sparse(means2d[vis], radii[vis], depths[vis], tmask, act, C, ts, tw, th,
                                             image_ids=ci)
        assert torch.equal(offp_r, offp_o) and torch.equal(flp_r, flp_o), (name, "packed isect")
        print(f"case {name}: C={C} N={N} {W}x{H} ts={ts} P={P}: AT={act.numel()} n_isects={fl_r.numel()} "
              f"(packed {flp_r.numel()})  oracle == reference")
        gold.update({f"{name}_dims": np.array([C, N, W, H, ts, P]), f"{name}_means2d": means2d.numpy(),
                     f"{name}_radii": radii.numpy(), f"{name}_depths": depths.numpy(), f"{name}_pixels": pixels.numpy(),
                     f"{name}_image_ids": image_ids.numpy(), f"{name}_active_tiles": lay_r[0].numpy(),
                     f"{name}_tile_mask": lay_r[1].numpy(), f"{name}_pixel_mask": u64(lay_r[2]),
                     f"{name}_pixel_cumsum": lay_r[3].numpy(), f"{name}_pixel_map": lay_r[4].numpy(),
                     f"{name}_tile_offsets": off_r.numpy(), f"{name}_flatten_ids": fl_r.numpy(),
                     f"{name}_tile_offsets_packed": offp_r.numpy(), f"{name}_flatten_ids_packed": flp_r.numpy()})
    np.savez_compressed(args.out, **gold)
    print(args.out, os.path.getsize(args.out) / 1e3, "kB")


if __name__ == "__main__":
    main()
