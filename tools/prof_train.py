import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import train_step_bench as T
from torch.profiler import profile, ProfilerActivity
# run the tool's loop under the profiler: monkeypatch time window by running run() with few steps
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    r = T.run(steps=20, refine=False)   # plain steps only
rows = [e for e in prof.key_averages() if e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.self_device_time_total)
print(r["ms_per_step"])
tot = sum(e.self_device_time_total for e in rows)
print("device total ms", tot/1e3)
for e in rows[:40]:
    print(f"{e.key[:95]:95s} n={e.count:5d} total_us={e.self_device_time_total:10.1f}")
