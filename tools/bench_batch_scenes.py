#!/usr/bin/env python
"""The reference's batched-scenes table (docs/batch.md:34-37: 10 k Gaussians, 300 x 200, 1 / 4 / 16 / 64 scenes per call,
3DGS forward and backward seconds) on this backend: one rasterization() call over batch dimension S, timed like theirs
(mean over repeated calls, synchronised). Prints one JSON row per S next to the published numbers."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsplat_amd

PUBLISHED = {1: (0.00037, 0.00049), 4: (0.00040, 0.00079), 16: (0.00093, 0.00284), 64: (0.00368, 0.01124)}
dev = torch.device("cuda", 0)
N, W, H = 10_000, 300, 200
g = torch.Generator(device=dev).manual_seed(0)
for S in (1, 4, 16, 64):
    means = torch.randn(S, N, 3, device=dev, generator=g) * 1.5 + torch.tensor([0.0, 0.0, 6.0], device=dev)
    quats = torch.nn.functional.normalize(torch.randn(S, N, 4, device=dev, generator=g), dim=-1)
    scales = torch.rand(S, N, 3, device=dev, generator=g) * 0.05 + 0.005
    opac = torch.rand(S, N, device=dev, generator=g)
    colors = torch.rand(S, N, 3, device=dev, generator=g)
    viewmats = torch.eye(4, device=dev).expand(S, 1, 4, 4).contiguous()
    Ks = torch.tensor([[300.0, 0, W / 2], [0, 300.0, H / 2], [0, 0, 1]], device=dev).expand(S, 1, 3, 3).contiguous()
    leaves = [t.requires_grad_(True) for t in (means, quats, scales, opac, colors)]

    def fwd():
        return gsplat_amd.rasterization(*leaves, viewmats, Ks, W, H)

    def timeit(f, n=50):
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n, out

    t_f, out = timeit(fwd)
    loss = out[0].sum()

    def bwd():
        loss.backward(retain_graph=True)
        for v in leaves:
            v.grad = None

    t_b, _ = timeit(bwd)
    print(json.dumps({"scenes": S, "gaussians_per_scene": N, "fwd_s": round(t_f, 6), "bwd_s": round(t_b, 6),
                      "published_fwd_bwd_s": PUBLISHED[S]}), flush=True)
