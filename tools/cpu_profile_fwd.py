#!/usr/bin/env python
"""cProfile of the host side of rasterization() FORWARD only on a tiny scene (GPU time negligible). GPU box."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, gsplat_amd
dev = torch.device("cuda", 0)
sc, W, H = bench.make_workload(2000, dev)
colors = torch.rand(2000, 3, device=dev)
def fwd():
    with torch.no_grad():
        return gsplat_amd.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], colors, sc["viewmats"], sc["Ks"], W, H)
def fwd_grad():
    return gsplat_amd.rasterization(sc["means"].requires_grad_(True), sc["quats"], sc["scales"], sc["opacities"], colors, sc["viewmats"], sc["Ks"], W, H)
for name, f in (("no_grad", fwd), ("grad", fwd_grad)):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): f()
    torch.cuda.synchronize(); print(name, "fwd ms/call:", round((time.perf_counter() - t0) / 300 * 1e3, 4))
pr = cProfile.Profile(); pr.enable()
for _ in range(300): fwd_grad()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(28)
