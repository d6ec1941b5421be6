#!/usr/bin/env python
"""torch.profiler table of one fwd+bwd step (kernels and the torch ops around them): prof_step.py [3dgs|2dgs] [packed]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, gsplat_amd
mode = sys.argv[1] if len(sys.argv) > 1 else "3dgs"
packed = "packed" in sys.argv
dev = torch.device("cuda", 0)
sc, W, H = bench.make_workload(1_000_000, dev)
names = ("means", "quats", "scales", "opacities", "colors")
leaves = {k: sc[k].clone().requires_grad_(True) for k in names}
def step():
    for t in leaves.values(): t.grad = None
    a = (leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"], sc["viewmats"], sc["Ks"], W, H)
    if mode == "2dgs":
        rc, ra, rn, sn, rd, rm, m = gsplat_amd.rasterization_2dgs(*a, sh_degree=3, packed=packed, render_mode="RGB+ED", distloss=True)
        (rc.sum() + rn.sum() + rd.sum()).backward()
    else:
        rc, ra, meta = gsplat_amd.rasterization(*a, sh_degree=3, packed=packed, tile_size=16)
        rc.sum().backward()
for _ in range(5): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=False) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print(f"total device time per step {tot / 3 / 1e3:.3f} ms")
for e in rows[:45]:
    print(f"{e.key[:100]:100s} n={e.count:4d} per_step_us={e.self_device_time_total / 3:9.1f}")
if "shapes" in sys.argv:
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof2:
        for _ in range(3): step()
        torch.cuda.synchronize()
    print("---- torch ops by shape (per step us)")
    for e in sorted(prof2.key_averages(group_by_input_shape=True), key=lambda e: -e.self_device_time_total):
        if e.key.startswith("aten::") and e.self_device_time_total / 3 > 3:
            print(f"{e.key:22s} n={e.count / 3:5.1f} us={e.self_device_time_total / 3:8.1f} {str(e.input_shapes)[:110]}")
