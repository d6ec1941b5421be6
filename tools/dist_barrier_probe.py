import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, gsplat_amd
import torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", RANK="0", WORLD_SIZE="1")
dist.init_process_group(backend="nccl", device_id=dev)
sc, W, H = bench.make_workload(1_000_000, dev)
names = ("means", "quats", "scales", "opacities", "colors")
leaves = {k: sc[k].clone().requires_grad_(True) for k in names}
def step():
    for t in leaves.values(): t.grad = None
    rc, ra, meta = gsplat_amd.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"],
                                            leaves["colors"], sc["viewmats"], sc["Ks"], W, H, sh_degree=3, packed=False, distributed=True)
    rc.sum().backward()
def T(): torch.cuda.synchronize(); return time.perf_counter()
for _ in range(5): step()
t0 = T(); dist.barrier(); t1 = T(); print("barrier A", round((t1 - t0) * 1e3, 3), flush=True)
for _ in range(20): step()
t2 = T(); print("20 steps", round((t2 - t1) * 1e3 / 20, 3), "ms/step", flush=True)
dist.barrier(); t3 = T(); print("barrier B", round((t3 - t2) * 1e3, 3), flush=True)
for _ in range(20): step()
t4 = time.perf_counter(); dist.barrier(); t5 = T(); print("20 steps + barrier without sync before", round((t5 - t3) * 1e3 / 20, 3), "ms/step; barrier call itself", round((t5 - t4) * 1e3, 3), flush=True)
for _ in range(20): step()
t6 = time.perf_counter(); x = torch.zeros(1, device=dev); dist.all_reduce(x); t7 = T(); print("20 steps + all_reduce barrier", round((t7 - t5) * 1e3 / 20, 3), "ms/step; all_reduce+sync", round((t7 - t6) * 1e3, 3), flush=True)
dist.destroy_process_group()
