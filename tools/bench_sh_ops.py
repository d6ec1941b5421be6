#!/usr/bin/env python
"""Times the SH ops at 1 M Gaussians / one camera (HIP-event stage times): full-band kernels (csrc/sh.hip) against the band
kernels behind spherical_harmonics_l1_plus and the fp16-coefficient path (csrc/sh_band.hip), with their share of 8 TB/s."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsplat_amd
from gsplat_amd import _cabi
dev = torch.device("cuda", 0)
N, K = 1_000_000, 16
g = torch.Generator().manual_seed(0)
means = torch.randn(N, 3, generator=g).to(dev)
viewmats = torch.eye(4, device=dev)[None].contiguous()
full = (torch.randn(N, K, 3, generator=g) * 0.3).to(dev)
masks = (torch.rand(1, N, generator=g) < 0.93).to(dev)

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); _cabi.profile_begin()
    for _ in range(n): fn()
    torch.cuda.synchronize(); p = _cabi.profile_end()
    return {k.replace("gsx_", ""): round(sum(v) / len(v) * 1e3, 1) for k, v in p.items()}

out = {}
for name, coeffs, fn in (
    ("full fp32 [N,16,3]", full, lambda c, m: gsplat_amd.spherical_harmonics(3, means, viewmats, c, masks=m)),
    ("l1_plus fp32 [N,15,3]", full[:, 1:].contiguous(), lambda c, m: gsplat_amd.spherical_harmonics_l1_plus(3, means, viewmats, c, masks=m)),
    ("full fp16 [N,16,3]", full.half(), lambda c, m: gsplat_amd.spherical_harmonics(3, means, viewmats, c, masks=m)),
):
    for with_means in (False, True):
        c = coeffs.clone().requires_grad_(True)
        mm = means.clone().requires_grad_(with_means)
        def step():
            nonlocal_means = mm
            col = (gsplat_amd.spherical_harmonics(3, nonlocal_means, viewmats, c, masks=masks) if "l1_plus" not in name
                   else gsplat_amd.spherical_harmonics_l1_plus(3, nonlocal_means, viewmats, c, masks=masks))
            col.sum().backward()
            c.grad = None; mm.grad = None
        out[f"{name} v_means={with_means}"] = timed(step)
esz = {"full fp32 [N,16,3]": 192, "l1_plus fp32 [N,15,3]": 180, "full fp16 [N,16,3]": 96}
for k, v in out.items():
    b = esz[k.split(" v_means")[0]]
    fwd_bytes, bwd_bytes = (b + 12 + 12 + 1) * N, (2 * b + 12 + 12 + 1 + (12 if "True" in k else 0)) * N
    ts = sorted(v.items())
    print(k, v, "| share of 8 TB/s:", [round((fwd_bytes if "fwd" in n else bwd_bytes) / (t * 1e-6) / 8e12, 3) for n, t in ts])
