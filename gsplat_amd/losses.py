"""Image losses of the training step around the rasterizer.

`ssim_loss` mirrors ``gsplat/losses.py:150-200`` (``1 - SSIM`` with an 11 x 11 Gaussian window of sigma 1.5, zero padding,
constants C1 = 0.01^2 / C2 = 0.03^2, mean over batch, channels and pixels - the term the reference trainer blends with L1,
``examples/simple_trainer.py:951-961``: ``loss = lerp(l1, ssim_loss, 0.2)``). The reference evaluates five depthwise 11 x 11
convolutions (or the third-party ``fused_ssim`` CUDA extension, not available here); the window is an outer product, so this
version runs the five maps as ONE stacked tensor through a vertical and a horizontal 11-tap pass - the same sums, 22 taps per
output instead of 121. Plain torch ops: differentiable through autograd, runs on any device.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor

_WINDOW_CACHE: dict = {}


def _window_1d(window_size: int, sigma: float, device, dtype) -> Tensor:
    key = (window_size, sigma, str(device), dtype)
    w = _WINDOW_CACHE.get(key)
    if w is None:
        x = torch.arange(window_size, device=device, dtype=torch.float32)
        g = torch.exp(-((x - window_size // 2) ** 2) / (2 * sigma ** 2))
        w = (g / g.sum()).to(dtype)
        _WINDOW_CACHE[key] = w
    return w


def ssim_map(img1: Tensor, img2: Tensor, window_size: int = 11) -> Tensor:
    """SSIM map [B, C, H, W] of two image batches [B, C, H, W] in [0, 1] (gsplat/losses.py: torch_ssim_loss)."""
    if img1.shape != img2.shape or img1.dim() != 4:
        raise ValueError(f"ssim: expected two [B, C, H, W] batches of one shape, got {tuple(img1.shape)} / {tuple(img2.shape)}")
    B, C, H, W = img1.shape
    w = _window_1d(window_size, 1.5, img1.device, img1.dtype)
    pad = window_size // 2
    maps = torch.cat([img1, img2, img1 * img1, img2 * img2, img1 * img2], dim=1)  # [B, 5 C, H, W]
    G = 5 * C
    maps = F.conv2d(maps, w.view(1, 1, window_size, 1).expand(G, 1, window_size, 1), padding=(pad, 0), groups=G)
    maps = F.conv2d(maps, w.view(1, 1, 1, window_size).expand(G, 1, 1, window_size), padding=(0, pad), groups=G)
    mu1, mu2, s11, s22, s12 = maps.split(C, dim=1)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    sigma1_sq, sigma2_sq, sigma12 = s11 - mu1_sq, s22 - mu2_sq, s12 - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))


class _FusedSsimMean(torch.autograd.Function):
    """mean(SSIM map) on the MI355X: csrc/ssim.hip (gsx_ssim_fwd / gsx_ssim_bwd), one kernel per direction, images read
    through their strides (a [B, H, W, C] render permuted to [B, C, H, W] is consumed in place). Differentiable in img1."""

    @staticmethod
    def forward(ctx, img1: Tensor, img2: Tensor):
        import ctypes

        from . import _cabi

        B, C, H, W = img1.shape
        img2 = img2.detach()
        dev = img1.device
        partial = torch.empty(_cabi._lib.gsx_ssim_blocks(B, C, H, W), device=dev, dtype=torch.float32)
        need_grad = ctx.needs_input_grad[0]
        dmaps = torch.empty((B, C, H, W, 3), device=dev, dtype=torch.float32) if need_grad else None
        s1, s2 = (ctypes.c_int64 * 4)(*img1.stride()), (ctypes.c_int64 * 4)(*img2.stride())
        _cabi.call("gsx_ssim_fwd", _cabi.ptr_strided(img1), s1, _cabi.ptr_strided(img2), s2, B, C, H, W, _cabi.ptr(partial),
                   _cabi.ptr(dmaps))
        ctx.save_for_backward(img1, img2, dmaps)
        return partial.sum() / float(B * C * H * W)

    @staticmethod
    @torch.autograd.function.once_differentiable  # the backward kernel is not itself differentiable
    def backward(ctx, v_mean: Tensor):
        import ctypes

        from . import _cabi

        img1, img2, dmaps = ctx.saved_tensors
        if dmaps is None:  # forward ran without a gradient request for img1
            return None, None
        B, C, H, W = img1.shape
        v_img1 = torch.empty_like(img1)  # preserve_format: a dense permuted view (channels-last render) keeps its strides
        s1, s2, sv = ((ctypes.c_int64 * 4)(*t.stride()) for t in (img1, img2, v_img1))
        v_mean = v_mean.reshape(1).to(torch.float32).contiguous()  # stays on the device: the kernel multiplies by it
        _cabi.call("gsx_ssim_bwd", _cabi.ptr_strided(img1), s1, _cabi.ptr_strided(img2), s2, B, C, H, W, _cabi.ptr(dmaps),
                   1.0 / float(B * C * H * W), _cabi.ptr(v_mean), _cabi.ptr_strided(v_img1), sv)
        return v_img1, None


def ssim_loss(img1: Tensor, img2: Tensor, window_size: int = 11) -> Tensor:
    """``1 - mean(SSIM)`` (gsplat/losses.py:150-200). float32 images on the GPU with the default window take the fused
    kernels (what the reference gets from the third-party ``fused_ssim`` extension); anything else the torch evaluation.
    Zero ("same") padding like gsplat/losses.py's torch path; the reference's optional fused_ssim fast path defaults to
    padding="valid" and is a different loss value - not offered here."""
    if (img1.is_cuda and img2.is_cuda and window_size == 11 and img1.dtype == torch.float32 and img2.dtype == torch.float32
            and img1.dim() == 4 and img1.shape == img2.shape and not img2.requires_grad
            and img1.shape[0] * img1.shape[1] <= 65535):  # the kernels' grid: one z-slice per (batch, channel) plane
        return 1.0 - _FusedSsimMean.apply(img1, img2)
    return 1.0 - ssim_map(img1, img2, window_size).mean()


def l1_loss(pred: Tensor, target: Tensor) -> Tensor:
    """Per-element L1 (gsplat/losses.py:48-63)."""
    return (pred - target).abs()
