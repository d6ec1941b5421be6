#!/usr/bin/env python
"""cProfile of the distributed=True step in a 1-rank RCCL group (host side). usage: dist_profile.py [n_gaussians] [n_cameras]"""
import cProfile, os, pstats, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import bench, gsplat_amd
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
c = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sc, W, H = bench.make_workload(n, dev, n_cameras=c)
leaves = {k: sc[k].clone().requires_grad_(True) for k in bench.NAMES}
def step():
    for t in leaves.values(): t.grad = None
    rc, ra, meta = gsplat_amd.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"],
                                            sc["viewmats"], sc["Ks"], W, H, sh_degree=3, packed=False, distributed=True)
    rc.sum().backward()
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per step")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
dist.destroy_process_group()
