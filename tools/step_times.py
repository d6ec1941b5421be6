#!/usr/bin/env python
"""Per-step device time of the first steps of the c3 bench step (HIP events around each step, no sync in between): is the first
timed window slower because of clocks (a gradual decline) or because of state that settles after N calls (a step change)?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, gsplat_amd

dev = torch.device("cuda", 0)
sc, W, H = bench.make_workload(1_000_000, dev, n_cameras=1)
leaves = {k: sc[k].clone().requires_grad_(True) for k in bench.NAMES}


def step():
    for t in leaves.values():
        t.grad = None
    rc, ra, meta = gsplat_amd.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"],
                                            leaves["colors"], sc["viewmats"], sc["Ks"], W, H, sh_degree=3, packed=False, tile_size=16)
    rc.sum().backward()


n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
torch.cuda.synchronize()
ev[0].record()
for i in range(n):
    step()
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
for i in range(0, n, 10):
    print(f"steps {i:3d}-{i + 9:3d}: " + " ".join(f"{x:.3f}" for x in ms[i:i + 10]))
