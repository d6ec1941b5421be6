"""CPU oracle of the whole rasterization() pipeline (TEST INFRASTRUCTURE ONLY — see oracle/oracle.py).

Chains the oracle stages exactly like the reference orchestrator (gsplat/cuda/csrc/Rendering.cpp:745-1481,
python restatement gsplat/rendering.py:722-1100): projection -> opacities (x compensation) -> SH (+0.5, clamp)
-> depth channel -> tile intersection (ellipse test) -> sort -> offsets -> compositing -> expected depth.
Gradients: analytic C backward for compositing, chained into torch autograd for the per-Gaussian stages.
Used by tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import math
import time
from typing import Dict, Optional

import torch

from . import oracle as O


def rasterization_cpu(
    means, quats, scales, opacities, colors, viewmats, Ks, width: int, height: int, sh_degree: Optional[int] = None,
    render_mode: str = "RGB", rasterize_mode: str = "classic", backgrounds=None, tile_size: int = 16,
    near_plane: float = 0.01, far_plane: float = 1e10, radius_clip: float = 0.0, eps2d: float = 0.3,
    camera_model: str = "pinhole", v_render_colors=None, v_render_alphas=None, want_grads: bool = True,
) -> Dict:
    """Single batch ([N,*] Gaussians, [C,*] cameras). Returns dict(render_colors, render_alphas, grads{...},
    n_isects, t_fwd, t_bwd). If cotangents are None, loss = render_colors.sum() (profiling/main.py:140-149)."""
    t0 = time.perf_counter()
    names = ("means", "quats", "scales", "opacities", "colors")
    leaf = {k: v.detach().clone().float().requires_grad_(want_grads)
            for k, v in zip(names, (means, quats, scales, opacities, colors))}
    C = viewmats.shape[0]
    calc_comp = rasterize_mode == "antialiased"
    rad, m2, dep, con, comp = O.fully_fused_projection(
        leaf["means"][None], None, leaf["quats"][None], leaf["scales"][None], viewmats[None], Ks[None], width, height,
        eps2d, near_plane, far_plane, radius_clip, calc_comp, camera_model, leaf["opacities"][None])
    rad, m2, dep, con = rad[0], m2[0], dep[0], con[0]
    op = leaf["opacities"][None].expand(C, -1)
    if comp is not None:
        op = op * comp[0]
    has_color = render_mode in ("RGB", "RGB+D", "RGB+ED")
    has_depth = render_mode in ("D", "ED", "RGB+D", "RGB+ED")
    feats = None
    if has_color:
        if sh_degree is None:
            feats = leaf["colors"][None].expand(C, -1, -1) if leaf["colors"].dim() == 2 else leaf["colors"]
        else:
            col = O.spherical_harmonics(sh_degree, leaf["means"][None], viewmats[None], leaf["colors"],
                                        (rad > 0).all(-1)[None])[0]
            feats = torch.clamp_min(col + 0.5, 0.0)
    bg = backgrounds
    if has_depth:
        feats = dep[..., None] if feats is None else torch.cat([feats, dep[..., None]], -1)
        if bg is not None:
            bg = torch.cat([bg, torch.zeros_like(bg[..., :1])], -1) if has_color else torch.zeros(C, 1)
    tw, th = math.ceil(width / tile_size), math.ceil(height / tile_size)
    _, ids, fl = O.isect_tiles(m2, rad, dep, tile_size, tw, th, conics=con, opacities=op)
    off = O.isect_offset_encode(ids, C, tw, th)
    rc_raw, ra, li = O.rasterize_to_pixels(m2, con, feats, op, width, height, tile_size, off, fl, backgrounds=bg)
    expected = render_mode in ("ED", "RGB+ED")
    a_ = ra.clamp_min(1e-10)
    rc = torch.cat([rc_raw[..., :-1], rc_raw[..., -1:] / a_], -1) if expected else rc_raw
    t1 = time.perf_counter()
    out = {"render_colors": rc, "render_alphas": ra, "n_isects": int(ids.numel()), "t_fwd": t1 - t0,
           "n_visible": int((rad > 0).all(-1).sum()), "meta": {"radii": rad, "means2d": m2, "depths": dep,
                                                                "conics": con, "isect_ids": ids, "flatten_ids": fl,
                                                                "isect_offsets": off, "last_ids": li}}
    if not want_grads:
        return out
    v_rc = torch.ones_like(rc) if v_render_colors is None else v_render_colors.float()
    v_ra = torch.zeros_like(ra) if v_render_alphas is None else v_render_alphas.float()
    if expected:
        v_feat = torch.cat([v_rc[..., :-1], v_rc[..., -1:] / a_], -1)
        v_alpha = v_ra - (v_rc[..., -1:] * rc_raw[..., -1:] / (a_ * a_)) * (ra > 1e-10)
    else:
        v_feat, v_alpha = v_rc, v_ra
    gr = O.rasterize_to_pixels_bwd(m2, con, feats, op, width, height, tile_size, off, fl, ra, li, v_feat, v_alpha,
                                   backgrounds=bg)

    def t(k, ref):
        return torch.from_numpy(gr[k]).to(ref.dtype).reshape(ref.shape)

    torch.autograd.backward([m2, con, feats, op], [t("v_means2d", m2), t("v_conics", con), t("v_colors", feats),
                                                  t("v_opacities", op)])
    out["grads"] = {k: leaf[k].grad for k in names}
    if backgrounds is not None:
        vb = torch.from_numpy(gr["v_backgrounds"]).float()
        out["grads"]["backgrounds"] = vb[..., :backgrounds.shape[-1]] if has_color else None
    out["t_bwd"] = time.perf_counter() - t1
    return out
