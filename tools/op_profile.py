#!/usr/bin/env python
"""torch.profiler table of one bench step (which torch op launches which glue kernel). Run on the GPU box."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, gsplat_amd
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
sc, W, H = bench.make_workload(1_000_000, dev)
names = ("means", "quats", "scales", "opacities", "colors")
leaves = {k: sc[k].clone().requires_grad_(True) for k in names}
packed = "packed" in sys.argv[1:]
distributed = "dist" in sys.argv[1:]
if distributed:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", device_id=dev)

def step():
    for t in leaves.values():
        t.grad = None
    rc, ra, meta = gsplat_amd.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"],
                                            leaves["colors"], sc["viewmats"], sc["Ks"], W, H, sh_degree=3, packed=packed,
                                            distributed=distributed)
    rc.sum().backward()

for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=70))
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.cpu_parent is None]
print("top-level CPU ops per step, self cpu us:")
import collections
agg = collections.Counter(); cnt = collections.Counter()
for e in ev:
    agg[e.name] += e.cpu_time_total; cnt[e.name] += 1
for k, v in agg.most_common(40):
    print(f"{k[:70]:70s} n={cnt[k]/3:5.1f} cpu_us/step={v/3:8.1f}")

print("copy-like ops by input shape (per step):")
shp = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name in ("aten::copy_", "aten::clone", "aten::contiguous",
                                                                     "aten::add", "aten::add_", "aten::zeros", "aten::fill_",
                                                                     "aten::zero_", "aten::sum", "aten::cat", "aten::index_select"):
        par = e.cpu_parent.name if e.cpu_parent is not None else "-"
        shp[(e.name, str(e.input_shapes)[:90], par[:50])] += 1
for (name, shapes, par), n in sorted(shp.items(), key=lambda kv: -kv[1]):
    print(f"  {name:18s} n={n/3:4.1f}  parent={par:50s} {shapes}")
