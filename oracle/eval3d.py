"""CPU oracle for the from-world ("eval3d") compositing of 3DGUT (TEST INFRASTRUCTURE ONLY; nothing under gsplat_amd/ may
import it). No product kernel exists for this stage yet — this is the checker it will be built against.

Restates ``gsplat::rasterize_to_pixels_from_world_3dgs`` (reference kernels
``gsplat/cuda/csrc/RasterizeToPixelsFromWorld3DGS*.cu``; the reference's torch statement is
``gsplat/cuda/_torch_impl_eval3d.py:135-495``): instead of evaluating a projected 2D conic at the pixel, every sample is
the response of the 3D Gaussian along the pixel's ray,

    M      = S^-1 R^T                               (``_compute_gaussian_transform`` :135-170)
    o', d' = M (ray_o - mean),  M ray_d / |M ray_d| (``_compute_ray_gaussian_distance`` :173-223)
    hit_t  = -d' . o'          (a Gaussian whose closest point lies behind the ray origin contributes nothing)
    dist2  = |d' x o'|^2
    alpha  = min(opacity * exp(-dist2 / 2), 1 - sqrt(1e-4))          (``_compute_gaussian_alphas`` :226-261)

followed by the same front-to-back compositing as the classic path: samples with alpha < 1/255 are skipped, the pixel stops
before the sample that would take the transmittance to <= 1e-4 (``accumulate_eval3d`` :264-495). Gradients come from torch
autograd on this vectorised statement, as the reference obtains its own.

Pinned: ``oracle/pin_eval3d_against_reference.py`` drives the reference's ``accumulate_eval3d`` (with a restated nerfacc,
as ``pin_against_reference.py`` does for the classic ``accumulate``) on the same candidate lists and writes
``tests/golden/eval3d_ref.npz``; ``tests/test_oracle_eval3d.py`` checks this module against those vectors.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
from torch import Tensor

ALPHA_THRESHOLD = 1.0 / 255.0
TRANSMITTANCE_THRESHOLD = 1e-4
MAX_ALPHA = 1.0 - math.sqrt(TRANSMITTANCE_THRESHOLD)


def _rotmat(q: Tensor) -> Tensor:
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(q.shape[:-1] + (3, 3))


def pinhole_rays(viewmats: Tensor, Ks: Tensor, width: int, height: int) -> Tensor:
    """World-space rays through the pixel centres of perfect pinhole cameras: [C, H, W, 6] = origin | unit direction
    (``_generate_rays`` :91-132 for the pinhole model: direction = R^T K^-1 (x + 0.5, y + 0.5, 1), origin = -R^T t)."""
    C = viewmats.shape[0]
    ys, xs = torch.meshgrid(torch.arange(height, dtype=viewmats.dtype), torch.arange(width, dtype=viewmats.dtype), indexing="ij")
    fx, fy, cx, cy = (Ks[:, 0, 0], Ks[:, 1, 1], Ks[:, 0, 2], Ks[:, 1, 2])
    d = torch.stack([(xs[None] + 0.5 - cx[:, None, None]) / fx[:, None, None],
                     (ys[None] + 0.5 - cy[:, None, None]) / fy[:, None, None],
                     torch.ones(C, height, width, dtype=viewmats.dtype)], dim=-1)
    d = d / d.norm(dim=-1, keepdim=True)
    R, t = viewmats[:, :3, :3], viewmats[:, :3, 3]
    d_w = torch.einsum("cji,chwj->chwi", R, d)
    o_w = -torch.einsum("cji,cj->ci", R, t)
    return torch.cat([o_w[:, None, None, :].expand(C, height, width, 3), d_w], dim=-1)


def candidate_lists(isect_offsets: Tensor, flatten_ids: Tensor, width: int, height: int, tile_size: int) -> Tuple[Tensor, Tensor]:
    """Per pixel, the depth-sorted rows of its tile, padded: (rows int64 [I*H*W, L], present bool [I*H*W, L])."""
    I, th, tw = isect_offsets.shape
    off = torch.cat([isect_offsets.reshape(-1).long(), torch.tensor([flatten_ids.numel()])])
    lens = off[1:] - off[:-1]
    L = int(lens.max()) if lens.numel() else 0
    ys, xs = torch.meshgrid(torch.arange(height), torch.arange(width), indexing="ij")
    tile = (ys // tile_size) * tw + (xs // tile_size)  # [H, W]
    tile = (torch.arange(I)[:, None, None] * (th * tw) + tile[None]).reshape(-1)  # [I*H*W]
    k = torch.arange(max(L, 1))
    present = k[None, :] < lens[tile][:, None]
    idx = (off[tile][:, None] + k[None, :]).clamp(max=max(flatten_ids.numel() - 1, 0))
    rows = flatten_ids.long()[idx] if flatten_ids.numel() else torch.zeros_like(idx)
    return rows, present


def rasterize_to_pixels_eval3d(
    means: Tensor, quats: Tensor, scales: Tensor, colors: Tensor, opacities: Tensor, rays: Tensor, image_width: int,
    image_height: int, tile_size: int, isect_offsets: Tensor, flatten_ids: Tensor, backgrounds: Optional[Tensor] = None,
):
    """means [N,3], quats [N,4], scales [N,3] (one batch), colors [I,N,D], opacities [I,N], rays [I,H,W,6],
    isect_offsets int32 [I,th,tw], flatten_ids int32 [M] (row = image * N + gaussian, depth-sorted per tile) ->
    (renders [I,H,W,D], alphas [I,H,W,1], last_ids int32 [I,H,W] = position in flatten_ids of the last sample, -1 if none).
    Differentiable in means / quats / scales / colors / opacities."""
    I, N, D = colors.shape
    P = I * image_height * image_width
    rows, present = candidate_lists(isect_offsets, flatten_ids, image_width, image_height, tile_size)  # [P, L]
    g = rows % N
    img = torch.arange(I).repeat_interleave(image_height * image_width)[:, None].expand_as(rows)
    # [N,3,3] = S^-1 R^T, assembled in float64 and rounded once, as the reference does (:158-165): the entries reach
    # 1 / min scale and every later product inherits their rounding
    M = (torch.diag_embed(1.0 / scales.double()) @ _rotmat(quats.double()).transpose(-1, -2)).to(scales.dtype)
    ro = rays.reshape(P, 6)[:, None, :3] - means[g]  # [P, L, 3]
    rd = rays.reshape(P, 6)[:, None, 3:].expand_as(ro)
    o = torch.einsum("plij,plj->pli", M[g], ro)
    d = torch.einsum("plij,plj->pli", M[g], rd)
    d = d / d.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    hit_t = -(d * o).sum(-1)
    cr = torch.linalg.cross(d, o)
    dist2 = (cr * cr).sum(-1)
    resp = torch.exp(-0.5 * dist2)
    alpha = torch.clamp(opacities[img, g] * resp, max=MAX_ALPHA)
    live = present & (hit_t >= 0.0) & (alpha >= ALPHA_THRESHOLD)
    a = torch.where(live, alpha, torch.zeros_like(alpha))
    T_before = torch.cumprod(torch.cat([torch.ones_like(a[:, :1]), 1.0 - a[:, :-1]], dim=1), dim=1)
    keep = live & (T_before * (1.0 - a) > TRANSMITTANCE_THRESHOLD)
    # a pixel STOPS at the first sample that fails the transmittance test: later samples do not contribute either
    stopped = torch.cumsum((live & ~keep).long(), dim=1) > 0
    keep = keep & ~stopped
    w = torch.where(keep, a * T_before, torch.zeros_like(a))
    renders = torch.einsum("pl,pld->pd", w, colors[img, g])
    alphas = w.sum(-1, keepdim=True)
    if backgrounds is not None:
        bg = backgrounds.repeat_interleave(image_height * image_width, dim=0)
        renders = renders + (1.0 - alphas) * bg
    # position in flatten_ids of the last kept sample
    I_, th, tw = isect_offsets.shape
    off = torch.cat([isect_offsets.reshape(-1).long(), torch.tensor([flatten_ids.numel()])])
    ys, xs = torch.meshgrid(torch.arange(image_height), torch.arange(image_width), indexing="ij")
    tile = ((ys // tile_size) * tw + (xs // tile_size))
    tile = (torch.arange(I)[:, None, None] * (th * tw) + tile[None]).reshape(-1)
    k = torch.arange(rows.shape[1])[None, :].expand_as(rows)
    last_k = torch.where(keep, k, torch.full_like(k, -1)).max(dim=1).values
    last_ids = torch.where(last_k >= 0, off[tile] + last_k, torch.full_like(last_k, -1)).to(torch.int32)
    shape = (I, image_height, image_width)
    return renders.reshape(shape + (D,)), alphas.reshape(shape + (1,)), last_ids.reshape(shape)
