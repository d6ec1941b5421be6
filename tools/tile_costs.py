#!/usr/bin/env python
"""Per-tile work of a compositing BACKWARD launch on a scene (run on the GPU box): list length against the entries up to the
tile's last contributor (what the backward walks: early termination cut the rest), and the makespan of greedy list scheduling of
those costs on the one-wave-per-tile kernel's slots (12 per CU x 256 CUs), longest-first, whole tiles against slices.
    python tools/tile_costs.py [garden GRID | c3]"""
import heapq, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gsplat_amd

dev = torch.device("cuda", 0)
what = sys.argv[1] if len(sys.argv) > 1 else "garden"
if what == "garden":
    import bench_reference_profile as brp
    grid = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    means, quats, scales, opac, colors, viewmats, Ks = brp.load_scene(grid, dev)
    kw = dict(near_plane=0.01, far_plane=100.0, radius_clip=3.0)
    args = (means, quats, scales, opac, colors[:, :3], viewmats[:1], Ks[:1], 1920, 1080)
else:
    import bench
    sc, W, H = bench.make_workload(1_000_000, dev)
    kw = dict(sh_degree=3)
    args = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["colors"], sc["viewmats"], sc["Ks"], W, H)
with torch.no_grad():
    rc, ra, meta = gsplat_amd.rasterization(*args, packed=False, **kw)
    m2, con, op = meta["means2d"], meta["conics"], meta["opacities"].contiguous()
    off, fl = meta["isect_offsets"].contiguous(), meta["flatten_ids"]
    col = torch.rand(m2.shape[:-1] + (3,), device=dev)
    _, _, _, last_ids = torch.ops.gsplat.rasterize_to_pixels_3dgs(m2, con, col, op, None, None, 1920, 1080, 16, off, fl, False, False)
M = fl.numel()
offs = off.reshape(-1).long()
ends = torch.cat([offs[1:], torch.tensor([M], device=dev)])
n = ends - offs
th, tw = off.shape[-2:]
li = last_ids.reshape(1080, 1920)
pad = torch.zeros(th * 16, tw * 16, dtype=li.dtype, device=dev)
pad[:1080, :1920] = li
last = pad.reshape(th, 16, tw, 16).permute(0, 2, 1, 3).reshape(th * tw, 256).max(1).values.long()
cost = (torch.minimum(ends, last + 1) - offs).clamp_min(0)
n, cost = n.cpu().float(), cost.cpu().float()
q = lambda x, p: float(x.quantile(p))
out = {"scene": " ".join(sys.argv[1:]) or "garden 5", "tiles": int(n.numel()), "n_isects": M,
       "list_len": {"mean": float(n.mean()), "p50": q(n, .5), "p99": q(n, .99), "max": float(n.max())},
       "cost_to_last_contributor": {"mean": float(cost.mean()), "p50": q(cost, .5), "p99": q(cost, .99), "max": float(cost.max()),
                                    "sum": float(cost.sum())}}
P = 12 * 256
def makespan(costs):
    h = [0.0] * P
    heapq.heapify(h)
    for c in sorted(costs, reverse=True):
        heapq.heappush(h, heapq.heappop(h) + c)
    return max(h)
c = (cost + 10.0).tolist()
ideal = sum(c) / P
out["slots"] = P
out["whole_tiles_longest_first_x_ideal"] = round(makespan(c) / ideal, 3)
out["largest_tile_x_ideal"] = round(max(c) / ideal, 3)
for seg in (256, 512, 1024):
    pieces = []
    for x in cost.tolist():
        k = int(x // seg)
        pieces += [seg + 10.0] * k + ([x - k * seg + 10.0] if x - k * seg > 0 or k == 0 else [])
    out[f"slices_{seg}_x_ideal"] = round(makespan(pieces) / (sum(pieces) / P), 3)
    out[f"slices_{seg}_pieces"] = len(pieces)
print(json.dumps(out))
