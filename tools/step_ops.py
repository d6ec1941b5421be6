#!/usr/bin/env python
"""Which torch-side kernels run inside one bench step (c3, dense), and which operator launches each: torch.profiler over
three steps, device-time per (operator, kernel). For finding copies / fills / adds that the op bodies could avoid."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, gsplat_amd

dev = torch.device("cuda", 0)
N_G, N_C = int(os.environ.get("STEP_OPS_N", 1_000_000)), int(os.environ.get("STEP_OPS_C", 1))  # c4: 4000000 / 4
sc, W, H = bench.make_workload(N_G, dev, n_cameras=N_C)
leaves = {k: sc[k].clone().requires_grad_(True) for k in bench.NAMES}


def step():
    for t in leaves.values():
        t.grad = None
    rc, ra, meta = gsplat_amd.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"],
                                            leaves["colors"], sc["viewmats"], sc["Ks"], W, H, sh_degree=3, packed=False,
                                            tile_size=16)
    rc.sum().backward()


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
