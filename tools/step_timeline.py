#!/usr/bin/env python
"""Kernel timeline of ONE step (start offset, duration, gap to the previous kernel) from torch.profiler's trace: where the GPU
idles inside a step (host round trips, allocator calls, index ops). usage: step_timeline.py [packed] [far=<far_plane>]"""
import json, os, sys, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, gsplat_amd

packed = "packed" in sys.argv
far = next((float(a.split("=")[1]) for a in sys.argv if a.startswith("far=")), 1e10)
dev = torch.device("cuda", 0)
sc, W, H = bench.make_workload(1_000_000, dev, n_cameras=1)
leaves = {k: sc[k].clone().requires_grad_(True) for k in bench.NAMES}


def step():
    for t in leaves.values():
        t.grad = None
    rc, ra, meta = gsplat_amd.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"],
                                            leaves["colors"], sc["viewmats"], sc["Ks"], W, H, sh_degree=3, packed=packed,
                                            tile_size=16, far_plane=far)
    rc.sum().backward()
    return meta


for _ in range(5):
    meta = step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(4):
        step()
    torch.cuda.synchronize()
path = os.path.join(tempfile.gettempdir(), "trace.json")
prof.export_chrome_trace(path)
ev = [e for e in json.load(open(path))["traceEvents"] if e.get("cat") in ("kernel", "gpu_memcpy", "gpu_memset") and "ts" in e]
ev.sort(key=lambda e: e["ts"])
# steps are separated by the projection kernel: take the third one
starts = [i for i, e in enumerate(ev) if "project" in e["name"] and ("fwd" in e["name"] or "count" in e["name"])]
firsts = [i for k, i in enumerate(starts) if k == 0 or i - starts[k - 1] > 3]
a, b = firsts[2], firsts[3]
t0 = ev[a]["ts"]
prev_end, busy = t0, 0.0
print(f"packed={packed} far_plane={far} rows={meta['gaussian_ids'].numel() if packed else 'dense'}")
for e in ev[a:b]:
    gap = e["ts"] - prev_end
    busy += e["dur"]
    print(f"{e['ts'] - t0:9.1f} +{e['dur']:7.1f}  gap {gap:6.1f}  {e['name'][:90]}")
    prev_end = max(prev_end, e["ts"] + e["dur"])
print(f"step span {ev[b]['ts'] - t0:.1f} us, kernels {busy:.1f} us, idle {ev[b]['ts'] - t0 - busy:.1f} us")
