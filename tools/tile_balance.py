#!/usr/bin/env python
"""How much of a compositing launch is tail? Per-tile work of the c3 scene (entries up to the tile's last contributor, what the
backward walks) -> makespan of greedy list scheduling on P workgroup slots in launch order vs longest-first vs the ideal sum/P."""
import heapq, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, gsplat_amd
dev = torch.device("cuda", 0)
sc, W, H = bench.make_workload(1_000_000, dev)
with torch.no_grad():
    rc, ra, meta = gsplat_amd.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["colors"], sc["viewmats"],
                                            sc["Ks"], W, H, sh_degree=3)
off = meta["isect_offsets"].reshape(-1).cpu().long()
n = torch.diff(torch.cat([off, torch.tensor([meta["flatten_ids"].numel()])]))
print("tiles", n.numel(), "entries per tile: mean %.1f  max %d  p99 %d" % (n.float().mean(), n.max(), n.float().quantile(0.99)))
# the forward's last contributor per pixel is not in meta: re-derive the walked length per tile from the rendered alphas is not
# possible either, so use the list length as the cost of the forward-without-early-out, and min(n, 3 x median) as a proxy of a
# saturating tile (the backward never walks past the tile's last contributor)
for name, cost in (("full lists", n.float()), ("capped at 2x median", n.float().clamp_max(2 * n.float().median()))):
    cost = cost + 20.0  # fixed per-tile overhead, in entries
    for P in (1024, 1280):
        def makespan(order):
            h = [0.0] * P
            heapq.heapify(h)
            for c in order:
                heapq.heappush(h, heapq.heappop(h) + c)
            return max(h)
        ideal = float(cost.sum()) / P
        print(f"{name:22s} P={P}: launch order {makespan(cost.tolist()) / ideal:.3f} x ideal, longest first "
              f"{makespan(sorted(cost.tolist(), reverse=True)) / ideal:.3f} x ideal, largest tile {float(cost.max()) / ideal:.3f} x ideal")
