#!/usr/bin/env python
"""c5 (BASELINE.json configs[4]): rasterization_2dgs fwd+bwd, 1M surfels, 1080p, RGB+ED + normals + distortion loss.
Prints ms/step, Mpixels/s and the per-stage C-ABI times (HIP events). Run on the GPU box."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, gsplat_amd
from gsplat_amd import _cabi

dev = torch.device("cuda", 0)
packed = "packed" in sys.argv[1:]
sc, W, H = bench.make_workload(1_000_000, dev)
names = ("means", "quats", "scales", "opacities", "colors")
leaves = {k: sc[k].clone().requires_grad_(True) for k in names}

def step():
    for t in leaves.values():
        t.grad = None
    out = gsplat_amd.rasterization_2dgs(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"],
                                        leaves["colors"], sc["viewmats"], sc["Ks"], W, H, sh_degree=3, packed=packed,
                                        render_mode="RGB+ED", distloss=True)
    rc, ra, rn, sn, rd, rm, meta = out
    (rc.sum() + rn.sum() + rd.sum()).backward()
    return meta

for _ in range(5):
    meta = step()
torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 20 * 1e3
_cabi.profile_begin()
for _ in range(5):
    step()
prof = _cabi.profile_end()
print(json.dumps({"workload": "c5: 2DGS 1M surfels, 1080p, RGB+ED, normals, distloss, fwd+bwd", "packed": packed,
                  "ms_per_step": round(ms, 4), "Mpixels_per_s": round(W * H / ms / 1e3, 2),
                  "n_isects": int(meta["isect_ids"].numel()),
                  "stage_ms_per_step": {k.replace("gsx_", ""): round(sum(v) / 5, 4) for k, v in sorted(prof.items())}}))
