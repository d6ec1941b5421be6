#!/usr/bin/env python
"""Host-side cost of one rasterization() fwd+bwd step: the same step as bench.py on a scene so small that the GPU is never
the bottleneck (2 k Gaussians, 64 x 64), timed without and with cProfile. Run on the GPU box."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, gsplat_amd
dev = torch.device("cuda", 0)
sc, _, _ = bench.make_workload(2000, dev)
W, H = 64, 64
sc["Ks"] = sc["Ks"].clone(); sc["Ks"][:, 0, 2] = 32; sc["Ks"][:, 1, 2] = 32; sc["Ks"][:, 0, 0] = 40; sc["Ks"][:, 1, 1] = 40
leaves = {k: sc[k].clone().requires_grad_(True) for k in bench.NAMES}
def step(packed=False):
    for t in leaves.values():
        t.grad = None
    rc, ra, meta = gsplat_amd.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"],
                                            leaves["colors"], sc["viewmats"], sc["Ks"], W, H, sh_degree=3, packed=packed)
    rc.sum().backward()
for packed in (False, True):
    for _ in range(20):
        step(packed)
    torch.cuda.synchronize()
    import gc; gc.collect(); gc.freeze()
    t0 = time.perf_counter()
    for _ in range(200):
        step(packed)
    torch.cuda.synchronize()
    print(f"packed={packed}: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per fwd+bwd step (host-bound scene)")
if "--profile" in sys.argv:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(100):
        step(False)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
