"""Pins gsplat_amd/losses.py:ssim_loss against the reference's own implementation (TEST INFRASTRUCTURE; needs /root/reference).

Imports gsplat/losses.py (ssim_loss -> torch_ssim_loss: five depthwise 11 x 11 convolutions; fused_ssim is not installed), runs
both on seeded images, checks value and gradient, and writes tests/golden/ssim_ref.npz (inputs + the reference's loss and
gradient) for tests/test_losses.py, which runs without a reference checkout.
usage: PYTHONPATH=/root/reference python oracle/pin_losses_against_reference.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.environ.get("GSPLAT_REFERENCE_PATH", "/root/reference"))


def main():
    from gsplat.losses import ssim_loss as ref_ssim  # the reference

    from gsplat_amd.losses import ssim_loss

    g = torch.Generator().manual_seed(11)
    out = {}
    for tag, (B, C, H, W) in (("a", (2, 3, 37, 53)), ("b", (1, 1, 16, 16)), ("c", (1, 3, 64, 96))):
        x = torch.rand(B, C, H, W, generator=g)
        y = (x + 0.1 * torch.randn(B, C, H, W, generator=g)).clamp(0, 1)
        xr = x.clone().requires_grad_(True)
        lr = ref_ssim(xr, y)
        lr.backward()
        xo = x.clone().requires_grad_(True)
        lo = ssim_loss(xo, y)
        lo.backward()
        assert abs(float(lr) - float(lo)) < 2e-6, (tag, float(lr), float(lo))
        assert float((xr.grad - xo.grad).abs().max()) < 1e-6 + 1e-4 * float(xr.grad.abs().max()), tag
        out.update({f"{tag}_x": x.numpy(), f"{tag}_y": y.numpy(), f"{tag}_loss": np.float32(lr.item()),
                    f"{tag}_grad": xr.grad.numpy()})
        print(tag, "ssim loss", float(lr), "ours", float(lo), "max |grad diff|", float((xr.grad - xo.grad).abs().max()))
    path = os.path.join(ROOT, "tests", "golden", "ssim_ref.npz")
    np.savez_compressed(path, **out)
    print("LOSSES PINNED ->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
