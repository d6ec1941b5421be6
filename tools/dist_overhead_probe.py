import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, gsplat_amd
import torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
sc, W, H = bench.make_workload(1_000_000, dev)
names = ("means", "quats", "scales", "opacities", "colors")
leaves = {k: sc[k].clone().requires_grad_(True) for k in names}
def step(distributed):
    for t in leaves.values(): t.grad = None
    rc, ra, meta = gsplat_amd.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"],
                                            leaves["colors"], sc["viewmats"], sc["Ks"], W, H, sh_degree=3, packed=False, distributed=distributed)
    rc.sum().backward()
def timeit(tag, distributed):
    for _ in range(5): step(distributed)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step(distributed)
    torch.cuda.synchronize(); print(tag, round((time.perf_counter() - t0) / 20 * 1e3, 3), "ms/step", flush=True)
timeit("no process group, local", False)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
mode = sys.argv[1] if len(sys.argv) > 1 else "devid"
if mode == "devid": dist.init_process_group(backend="nccl", device_id=dev)
else: dist.init_process_group(backend="nccl")
timeit("process group up, local path", False)
timeit("process group up, distributed path", True)
dist.barrier(); torch.cuda.synchronize()
timeit("after a barrier, local path", False)
timeit("after a barrier, distributed path", True)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dist.barrier(); torch.cuda.synchronize()
    print("dist.barrier()", i, round((time.perf_counter() - t0) * 1e3, 3), "ms", flush=True)
x = torch.zeros(1, device=dev)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dist.all_reduce(x); torch.cuda.synchronize()
    print("all_reduce(1 elem) + sync", i, round((time.perf_counter() - t0) * 1e3, 3), "ms", flush=True)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dist.barrier(device_ids=[0]); torch.cuda.synchronize()
    print("dist.barrier(device_ids)", i, round((time.perf_counter() - t0) * 1e3, 3), "ms", flush=True)
dist.destroy_process_group()
