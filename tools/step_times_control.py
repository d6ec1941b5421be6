#!/usr/bin/env python
"""Controls for tools/step_times.py: the same per-step timing from process start for (a) an HBM-bound loop of elementwise kernels
and (b) a compute-bound loop (a chain of sin() over an L2-resident array: vector ALU, no HBM traffic)."""
import torch

dev = torch.device("cuda", 0)


def run(tag, step, n=80):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(n):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    for i in range(0, n, 10):
        print(f"{tag} {i:3d}: " + " ".join(f"{x:.3f}" for x in ms[i:i + 10]))


import sys
which = sys.argv[1] if len(sys.argv) > 1 else "valu"
if which == "hbm":
    a = torch.randn(64 * 1024 * 1024, device=dev)
    b = torch.empty_like(a)
    run("hbm", lambda: [torch.mul(a, 1.0001, out=b) for _ in range(6)])
else:
    x = torch.randn(256 * 4096, device=dev)  # 4 MB: stays in L2
    y = torch.empty_like(x)

    def valu():
        for _ in range(40):
            torch.sin(x, out=y)
    run("valu", valu)
