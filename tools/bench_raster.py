#!/usr/bin/env python
"""Kernel-level A/B harness for the compositing kernels (run on the GPU box).

Builds the bench.py c3 workload once, runs the pipeline up to the sorted tile lists, then times
gsx_raster3d_fwd / gsx_raster3d_bwd alone with HIP events (n repeats). Select a library variant with
GSPLAT_AMD_LIB=/path/to/libvariant.so (see gsplat_amd/_cabi.py) to compare builds in one gpurun call.
"""
import argparse
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--channels", type=int, default=3)
    ap.add_argument("--scale-mult", type=float, default=1.0)
    ap.add_argument("--absgrad", action="store_true")
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    import bench
    import gsplat_amd
    from gsplat_amd import _cabi

    dev = torch.device("cuda", 0)
    sc, W, H = bench.make_workload(args.gaussians, dev)
    if args.scale_mult != 1.0:
        sc["scales"] = sc["scales"] * args.scale_mult
    with torch.no_grad():
        rc, ra, meta = gsplat_amd.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["colors"],
                                                sc["viewmats"], sc["Ks"], W, H, sh_degree=3, packed=True)
    means2d, conics, opac = meta["means2d"], meta["conics"], meta["opacities"].contiguous()
    D = args.channels
    colors = torch.rand(means2d.shape[0], D, device=dev)
    offs, fl = meta["isect_offsets"].contiguous(), meta["flatten_ids"]
    ops = torch.ops.gsplat

    def fwd():
        return ops.rasterize_to_pixels_3dgs(means2d, conics, colors, opac, None, None, W, H, 16, offs, fl, True, False)

    renders, alphas, _, last_ids = fwd()
    v_rc = torch.ones_like(renders)
    v_ra = torch.zeros_like(alphas)

    def bwd():
        return ops.rasterize_to_pixels_3dgs_bwd(means2d, conics, colors, opac, None, None, offs, fl, alphas, last_ids,
                                                W, H, 16, args.absgrad, v_rc, v_ra, False)

    res = {"tag": args.tag, "lib": os.path.basename(_cabi.lib_path()), "n_isects": int(fl.numel()), "channels": D,
           "absgrad": args.absgrad, "env": {k: v for k, v in os.environ.items() if k.startswith("GSX_")}}
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(3):
            fn()
        _cabi.profile_begin()
        for _ in range(args.reps):
            fn()
        prof = _cabi.profile_end()
        (k, v), = [(k, v) for k, v in prof.items() if "raster" in k]
        v = sorted(v)
        res[name + "_us_median"] = round(1e3 * v[len(v) // 2], 1)
        res[name + "_us_min"] = round(1e3 * v[0], 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
