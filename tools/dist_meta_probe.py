#!/usr/bin/env python
"""GPU-box probe for the control plane of distributed rasterization: an RCCL world (size 1 on the 1-GPU box) next to
the gloo side group that carries the per-call (N_i, C_i) exchange. Checks that the gloo group can be created on the box
(interface resolution) and that a gather over it does not touch the device stream."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
from gsplat_amd import distributed as D  # noqa: E402

t0 = time.perf_counter()
grp = D._meta_group()
print("gloo side group:", grp, f"created in {(time.perf_counter() - t0) * 1e3:.1f} ms, ifname",
      os.environ.get("GLOO_SOCKET_IFNAME"))
assert grp is not None, "gloo side group could not be created"
mine = torch.tensor([123, 4], dtype=torch.int64)
out = [torch.empty(2, dtype=torch.int64)]
t0 = time.perf_counter()
for _ in range(100):
    dist.all_gather(out, mine, group=grp)
print("gather over gloo:", out[0].tolist(), f"{(time.perf_counter() - t0) * 10:.3f} ms per call")
x = torch.ones(4, device="cuda")
dist.all_reduce(x)
print("rccl all_reduce ok:", x.tolist())
dist.destroy_process_group()
