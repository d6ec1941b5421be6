#!/usr/bin/env python
"""Tile intersection alone on the c3 (default) or c4 scene: per-kernel-group HIP-event times of isect_tiles(). For ablation
builds whose outputs are not meant to be consumed (GSPLAT_AMD_LIB=...). usage: bench_isect.py [c4]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, gsplat_amd
from gsplat_amd import _cabi

dev = torch.device("cuda", 0)
c4 = "c4" in sys.argv[1:]
sc, W, H = bench.make_workload(4_000_000 if c4 else 1_000_000, dev, n_cameras=4 if c4 else 1)
with torch.no_grad():
    radii, means2d, depths, conics, _ = gsplat_amd.fully_fused_projection(
        sc["means"], None, sc["quats"], sc["scales"], sc["viewmats"], sc["Ks"], W, H, opacities=sc["opacities"])
    C = sc["viewmats"].shape[0]
    opac = sc["opacities"][None].expand(C, -1).contiguous()
    tw, th = (W + 15) // 16, (H + 15) // 16
    from gsplat_amd._wrapper import isect_tiles_begin, isect_tiles_finish
    def run():
        p = isect_tiles_begin(means2d, radii, depths, 16, tw, th, n_images=C, conics=conics, opacities=opac)
        return isect_tiles_finish(p)
    for _ in range(3):
        out = run()
    torch.cuda.synchronize()
    _cabi.profile_begin()
    for _ in range(10):
        out = run()
    prof = _cabi.profile_end()
print(json.dumps({"n_isects": int(out[1].numel()), **{k.replace("gsx_", ""): round(sum(v) / len(v), 4) for k, v in sorted(prof.items())}}))
