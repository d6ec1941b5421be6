"""Optimizers of the training step around the rasterizer (reference ``gsplat/optimizers/selective_adam.py``).

``SelectiveAdam`` = Adam restricted to the Gaussians that were visible in the step (Taming-3DGS), one fused HIP launch
per parameter tensor (C-ABI ``gsx_adam``, op ``torch.ops.gsplat.adam``). Like the reference kernel it applies NO bias
correction, and rows whose mask is False keep their parameter and both moments untouched.
"""
from __future__ import annotations

import torch

from . import _ops  # noqa: F401  (defines torch.ops.gsplat.adam)


class SelectiveAdam(torch.optim.Adam):
    """``step(visibility)``: ``visibility`` is a bool tensor with one entry per Gaussian (row of every parameter)."""

    def __init__(self, params, eps, betas):
        super().__init__(params=params, eps=eps, betas=betas)

    @torch.no_grad()
    def step(self, visibility):
        for group in self.param_groups:
            assert len(group["params"]) == 1, "SelectiveAdam expects one tensor per parameter group"
            param = group["params"][0]
            if param.grad is None:
                continue
            state = self.state[param]
            if len(state) == 0:  # lazy state, same keys as torch.optim.Adam
                state["step"] = torch.tensor(0.0, dtype=torch.float32)
                state["exp_avg"] = torch.zeros_like(param, memory_format=torch.preserve_format)
                state["exp_avg_sq"] = torch.zeros_like(param, memory_format=torch.preserve_format)
            beta1, beta2 = group["betas"]
            torch.ops.gsplat.adam(param, param.grad, state["exp_avg"], state["exp_avg_sq"],
                                  visibility.to(torch.bool), group["lr"], beta1, beta2, group["eps"])
