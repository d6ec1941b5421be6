"""MCMC relocation (reference ``gsplat/relocation.py:23-67``; 3DGS-as-MCMC, arXiv 2404.09591, Eq. 9)."""
from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor

from . import _ops  # noqa: F401


def compute_relocation(opacities: Tensor, scales: Tensor, ratios: Tensor, binoms: Tensor,
                       min_opacity: float = 0.005) -> Tuple[Tensor, Tensor]:
    """New opacities [N] and scales [N, 3] of Gaussians that are each split into ``ratios[i]`` copies.
    ``binoms`` [n_max, n_max] is the table of binomial coefficients; ``ratios`` is clamped to [1, n_max] in place,
    like the reference."""
    n = opacities.shape[0]
    n_max = binoms.shape[0]
    assert scales.shape == (n, 3), scales.shape
    assert ratios.shape == (n,), ratios.shape
    ratios.clamp_(min=1, max=n_max)
    return torch.ops.gsplat.relocation(opacities.contiguous(), scales.contiguous(), ratios.int().contiguous(), binoms,
                                       n_max, min_opacity)
