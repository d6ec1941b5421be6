"""Ordering of the splats on the square parameter grid before PNG compression.

The reference sorts with PLAS, *Parallel Linear Assignment Sorting* (``plas.sort_with_plas``, github.com/fraunhoferhhi/PLAS,
un-pinned in the reference; gsplat/compression/sort.py:22-64), a third-party package that is not part of this image. Any
permutation is a valid input of the codec (decompression never needs it back), the ordering only decides how well the PNG
filters predict a pixel from its neighbours. ``sort_splats`` therefore uses PLAS when it is importable - same call, same
keys as the reference - and otherwise lays the splats out along a space-filling curve: splats ordered by the Morton code of
their (log-transformed) positions are written along the Z-order traversal of the grid cells, so that splats close in space
become pixels close on the grid. It has the reference's contract (a perfect-square count, every field permuted alike) and
needs nothing beyond torch; on smooth scenes it recovers most of, not all of, the gain of the assignment search.
"""
from __future__ import annotations

from typing import Dict

import torch
from torch import Tensor


def _spread_bits(v: Tensor, stride: int, bits: int) -> Tensor:
    """bit i of v -> bit i * stride (int64)."""
    out = torch.zeros_like(v)
    for i in range(bits):
        out |= ((v >> i) & 1) << (i * stride)
    return out


def morton_order_3d(points: Tensor, bits: int = 16) -> Tensor:
    """Permutation that sorts [N, 3] points by the Morton code of their coordinates quantised to `bits` bits per axis."""
    lo, hi = points.amin(0), points.amax(0)
    q = ((points - lo) / (hi - lo).clamp_min(1e-30) * (2**bits - 1)).round().to(torch.int64).clamp_(0, 2**bits - 1)
    code = _spread_bits(q[:, 0], 3, bits) | (_spread_bits(q[:, 1], 3, bits) << 1) | (_spread_bits(q[:, 2], 3, bits) << 2)
    return torch.argsort(code, stable=True)


def z_curve_cells(side: int, device=None) -> Tensor:
    """The cells of a side x side grid (flat row-major indices) in Z-order."""
    ys, xs = torch.meshgrid(torch.arange(side, device=device), torch.arange(side, device=device), indexing="ij")
    bits = max(1, (side - 1).bit_length())
    code = _spread_bits(xs.reshape(-1), 2, bits) | (_spread_bits(ys.reshape(-1), 2, bits) << 1)
    return torch.argsort(code, stable=True)


def sort_splats(splats: Dict[str, Tensor], verbose: bool = True) -> Dict[str, Tensor]:
    """Reorder every field of `splats` (in place, and returned) for the square grid (reference sort.py:22)."""
    n = len(splats["means"])
    side = int(n**0.5)
    assert side * side == n, "Must be a perfect square"
    try:
        from plas import sort_with_plas
    except ImportError:
        sort_with_plas = None
    if sort_with_plas is not None:
        keys = ["means", "quats", "scales", "opacities"] + (["sh0"] if "sh0" in splats else [])
        feats = torch.cat([splats[k].reshape(n, -1) for k in keys], dim=-1)
        shuffle = torch.randperm(n, device=feats.device)
        grid = feats[shuffle].reshape(side, side, -1)
        _, idx = sort_with_plas(grid.permute(2, 0, 1), improvement_break=1e-4, verbose=verbose)
        order = shuffle[idx.squeeze().flatten()]
    else:
        means = splats["means"].detach().float()
        by_position = morton_order_3d(means)
        order = torch.empty(n, dtype=torch.int64, device=means.device)
        order[z_curve_cells(side, means.device)] = by_position
    for k, v in splats.items():
        splats[k] = v[order.to(v.device)]
    return splats
