import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench, gsplat_amd
dev = torch.device("cuda", 0)
sc, W, H = bench.make_workload(1_000_000, dev)
names = ("means", "quats", "scales", "opacities", "colors")
leaves = {k: sc[k].clone().requires_grad_(True) for k in names}
def step():
    for t in leaves.values(): t.grad = None
    out = gsplat_amd.rasterization_2dgs(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"],
                                        sc["viewmats"], sc["Ks"], W, H, sh_degree=3, packed=False, render_mode="RGB+ED", distloss=True)
    rc, ra, rn, sn, rd, rm, meta = out
    (rc.sum() + rn.sum() + rd.sum()).backward()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
want = ("aten::fill_", "aten::copy_", "aten::cat", "aten::zeros", "aten::zero_", "aten::add_", "aten::add", "aten::contiguous", "aten::clone", "aten::index_select", "aten::mul", "aten::sum")
for e in prof.events():
    if e.name in want and e.device_time_total > 3:
        st = [s for s in (e.stack or []) if "gsplat_amd" in s or "bench" in s or "stk.py" in s][:3]
        print(f"{e.device_time_total:8.1f} us {e.name:18s} {str(e.input_shapes)[:60]:60s} {' <- '.join(s.split('/')[-1][:60] for s in st)}")
