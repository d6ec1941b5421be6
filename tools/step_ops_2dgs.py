#!/usr/bin/env python
"""Which torch-side kernels run inside one c5 step (rasterization_2dgs, RGB+ED, normals, distortion), and which operator launches
each: torch.profiler over three steps, device time per (operator, kernel). The 2DGS counterpart of tools/step_ops.py."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, gsplat_amd

dev = torch.device("cuda", 0)
sc, W, H = bench.make_workload(1_000_000, dev)
names = ("means", "quats", "scales", "opacities", "colors")
leaves = {k: sc[k].clone().requires_grad_(True) for k in names}


def step():
    for t in leaves.values():
        t.grad = None
    out = gsplat_amd.rasterization_2dgs(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"],
                                        sc["viewmats"], sc["Ks"], W, H, sh_degree=3, packed=False, render_mode="RGB+ED", distloss=True)
    rc, ra, rn, sn, rd, rm, meta = out
    (rc.sum() + rn.sum() + rd.sum()).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = getattr(e, "self_cuda_time_total", 0)
    if dt > 0:
        rows.append((dt / 3, e.key, e.count // 3 or e.count, str(e.input_shapes)[:110]))
for dt, key, n, shapes in sorted(rows, reverse=True):
    print(f"{dt:9.1f} us/step  x{n:<3d} {key[:70]:70s} {shapes}")
