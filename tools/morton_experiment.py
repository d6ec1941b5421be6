#!/usr/bin/env python
"""What does the ORDER of the Gaussians in memory cost? Times a whole fwd+bwd step and the intersection alone on c3 / c4 with
the Gaussians in their random order and re-ordered along a 3-D Morton curve (what a trainer can keep up at every refinement:
`gsplat_amd.strategy.ops.morton_order`). Same Gaussians, same render.  usage: morton_experiment.py [c4]"""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, gsplat_amd
from gsplat_amd import _cabi

dev = torch.device("cuda", 0)
c4 = "c4" in sys.argv
sc, W, H = bench.make_workload(4_000_000 if c4 else 1_000_000, dev, n_cameras=4 if c4 else 1)


def morton3(p, bits=10):
    lo, hi = p.min(0).values, p.max(0).values
    q = ((p - lo) / (hi - lo + 1e-9) * ((1 << bits) - 1)).long()
    code = torch.zeros(p.shape[0], dtype=torch.long, device=p.device)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return code


def run(s, label):
    names = ("means", "quats", "scales", "opacities", "colors")
    leaves = {k: s[k].clone().requires_grad_(True) for k in names}

    def step():
        for t in leaves.values():
            t.grad = None
        rc, ra, meta = gsplat_amd.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"],
                                                leaves["colors"], s["viewmats"], s["Ks"], W, H, sh_degree=3, packed=False)
        rc.sum().backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    _cabi.profile_begin()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    prof = _cabi.profile_end()
    st = {k.replace("gsx_", ""): round(sum(v) / 5, 4) for k, v in sorted(prof.items())}
    print(json.dumps({"scene": "c4" if c4 else "c3", "order": label, "ms_per_step": round(ms, 4), "stages_ms": st}), flush=True)


run(sc, "random")
perm = torch.argsort(morton3(sc["means"]))
sorted_sc = dict(sc)
for k in ("means", "quats", "scales", "opacities", "colors"):
    sorted_sc[k] = sc[k][perm].contiguous()
run(sorted_sc, "morton")
