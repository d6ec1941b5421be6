"""CPU: gsplat_amd/losses.py against values produced by the REFERENCE's gsplat/losses.py (tests/golden/ssim_ref.npz, written by
oracle/pin_losses_against_reference.py) and against a direct restatement of the dense 11 x 11 window."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gsplat_amd.losses import l1_loss, ssim_loss, ssim_map

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ssim_loss_matches_reference_outputs():
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "ssim_ref.npz")))
    for tag in ("a", "b", "c"):
        x = torch.from_numpy(g[f"{tag}_x"]).requires_grad_(True)
        y = torch.from_numpy(g[f"{tag}_y"])
        loss = ssim_loss(x, y)
        loss.backward()
        assert abs(float(loss) - float(g[f"{tag}_loss"])) < 2e-6, tag
        ref = torch.from_numpy(g[f"{tag}_grad"])
        assert float((x.grad - ref).abs().max()) <= 1e-7 + 1e-4 * float(ref.abs().max()), tag


def test_ssim_map_equals_the_dense_window():
    """Wang et al. 2004 with the 2-D window w w^T evaluated as ONE 121-tap depthwise convolution per map (what
    gsplat/losses.py: torch_ssim_loss does) - the separable evaluation must give the same map."""
    g = torch.Generator().manual_seed(3)
    x, y = torch.rand(1, 3, 40, 56, generator=g), torch.rand(1, 3, 40, 56, generator=g)
    t = torch.arange(11, dtype=torch.float32)
    w1 = torch.exp(-((t - 5) ** 2) / (2 * 1.5 ** 2))
    w1 = w1 / w1.sum()
    w2 = (w1[:, None] * w1[None, :])[None, None].expand(3, 1, 11, 11).contiguous()
    conv = lambda a: F.conv2d(a, w2, padding=5, groups=3)  # noqa: E731
    mu1, mu2 = conv(x), conv(y)
    s1, s2, s12 = conv(x * x) - mu1 * mu1, conv(y * y) - mu2 * mu2, conv(x * y) - mu1 * mu2
    dense = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
    assert torch.allclose(ssim_map(x, y), dense, rtol=1e-4, atol=2e-5)
    assert float(ssim_loss(x, x)) < 1e-6  # identical images: SSIM = 1
    assert torch.equal(l1_loss(x, y), (x - y).abs())


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["nchw", "nhwc_view"])
def test_fused_ssim_kernels_match_the_reference_outputs_and_the_torch_evaluation(layout):
    """csrc/ssim.hip on the MI355X: the loss and its gradient against the values the REFERENCE's gsplat/losses.py produced
    (tests/golden/ssim_ref.npz) and, on a 1080p render-shaped pair read through channels-last strides, against the torch
    evaluation of the same formula."""
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "ssim_ref.npz")))

    def as_layout(t):
        if layout == "nchw":
            return t.cuda()
        return t.permute(0, 2, 3, 1).contiguous().cuda().permute(0, 3, 1, 2)  # [B, H, W, C] storage viewed as [B, C, H, W]

    for tag in ("a", "b", "c"):
        x = as_layout(torch.from_numpy(g[f"{tag}_x"])).requires_grad_(True)
        y = as_layout(torch.from_numpy(g[f"{tag}_y"]))
        loss = ssim_loss(x, y)
        loss.backward()
        assert abs(float(loss) - float(g[f"{tag}_loss"])) < 5e-6, (tag, float(loss), float(g[f"{tag}_loss"]))
        ref = torch.from_numpy(g[f"{tag}_grad"])
        assert float((x.grad.cpu() - ref).abs().max()) <= 2e-7 + 2e-4 * float(ref.abs().max()), tag
    gen = torch.Generator().manual_seed(5)
    img = torch.rand(1, 1080, 1920, 3, generator=gen).cuda()
    tgt = (img + 0.05 * torch.randn(1, 1080, 1920, 3, generator=gen).cuda()).clamp(0, 1)
    a = img.clone().requires_grad_(True)
    b = img.clone().requires_grad_(True)
    la = ssim_loss(a.permute(0, 3, 1, 2), tgt.permute(0, 3, 1, 2))          # fused kernels, strided reads
    lb = 1.0 - ssim_map(b.permute(0, 3, 1, 2), tgt.permute(0, 3, 1, 2)).mean()  # torch convolutions
    (3.0 * la).backward()
    (3.0 * lb).backward()
    assert abs(float(la) - float(lb)) < 2e-6
    assert float((a.grad - b.grad).abs().max()) <= 1e-9 + 1e-3 * float(b.grad.abs().max())
