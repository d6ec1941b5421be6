#!/usr/bin/env python
"""cProfile of the host side of one rasterization() fwd+bwd step on a tiny scene (GPU time negligible): where the
Python / dispatcher time of the shim goes. Run on the GPU box."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, gsplat_amd

dev = torch.device("cuda", 0)
sc, W, H = bench.make_workload(2000, dev)
names = ("means", "quats", "scales", "opacities", "colors")
leaves = {k: sc[k].clone().requires_grad_(True) for k in names}
packed = "packed" in sys.argv[1:]

def step():
    for t in leaves.values():
        t.grad = None
    rc, ra, meta = gsplat_amd.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"],
                                            leaves["colors"], sc["viewmats"], sc["Ks"], W, H, sh_degree=3, packed=packed)
    rc.sum().backward()

for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
torch.cuda.synchronize()
print("ms/step (tiny scene):", round((time.perf_counter() - t0) / 200 * 1e3, 4))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
