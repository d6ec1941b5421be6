import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import train_step_bench as T
from torch.profiler import profile, ProfilerActivity
# run the tool's loop under the profiler: monkeypatch time window by running run() with few steps
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    r = T.run(steps=20, refine=False, split_sh="split" in sys.argv)   # plain steps only
rows = [e for e in prof.key_averages() if e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.self_device_time_total)
print(r["ms_per_step"])
tot = sum(e.self_device_time_total for e in rows)
print("device total ms", tot/1e3)
for e in rows[:40]:
    print(f"{e.key[:95]:95s} n={e.count:5d} total_us={e.self_device_time_total:10.1f}")

print("---- copies / muls by shape")
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::copy_", "aten::mul", "aten::contiguous", "aten::clone") and e.self_device_time_total > 200:
        print(e.key, e.count, round(e.self_device_time_total, 1), str(e.input_shapes)[:120])
