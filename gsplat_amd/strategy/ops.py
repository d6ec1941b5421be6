"""In-place edits of a Gaussian set and of the optimizer state that goes with it (reference
``gsplat/strategy/ops.py``): duplicate / split / remove / reset_opa (DefaultStrategy) and relocate / sample_add /
inject_noise_to_position (MCMCStrategy). Same call signatures and semantics as the reference; parameters are stored
the way its trainer stores them (``scales`` = log-scales, ``opacities`` = logits, ``quats`` un-normalised wxyz).

Every edit is expressed as one row map — "new row j comes from old row src[j], fresh or inherited" — applied uniformly
to each parameter, to its optimizer moments (fresh rows start at zero) and to the per-Gaussian running statistics.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Union

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

from ..relocation import compute_relocation

Params = Union[Dict[str, torch.nn.Parameter], torch.nn.ParameterDict]


def _quat_to_rotmat(q: Tensor) -> Tensor:
    """Unit quaternions wxyz [..., 4] -> rotation matrices [..., 3, 3]."""
    w, x, y, z = q.unbind(-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y),
    ], dim=-1).reshape(q.shape[:-1] + (3, 3))


@torch.no_grad()
def _multinomial_sample(weights: Tensor, n: int, replacement: bool = True) -> Tensor:
    """torch.multinomial is limited to 2^24 categories; larger sets are sampled with numpy (reference ops.py:67-93)."""
    if weights.numel() <= 2 ** 24:
        return torch.multinomial(weights, n, replacement=replacement)
    p = (weights / weights.sum()).detach().cpu().numpy()
    idx = np.random.choice(weights.numel(), size=n, p=p, replace=replacement)
    return torch.from_numpy(idx).to(weights.device)


@torch.no_grad()
def _update_param_with_optimizer(param_fn: Callable[[str, Tensor], Tensor], optimizer_fn: Callable[[str, Tensor], Tensor],
                                 params: Params, optimizers: Dict[str, torch.optim.Optimizer],
                                 names: Optional[List[str]] = None):
    """Replace ``params[name]`` by ``param_fn(name, old)`` and move the optimizer state (every entry except "step"
    goes through ``optimizer_fn``) from the old tensor to the new one (reference ops.py:96-139)."""
    for name in (list(params.keys()) if names is None else names):
        old = params[name]
        new = param_fn(name, old)
        params[name] = new
        if name not in optimizers:
            assert not old.requires_grad, f"parameter '{name}' is trainable but has no optimizer"
            continue
        opt = optimizers[name]
        for group in opt.param_groups:
            st = opt.state.pop(old, {})
            for key in list(st.keys()):
                if key != "step":
                    st[key] = optimizer_fn(key, st[key])
            group["params"] = [new]
            opt.state[new] = st


def _as_param(t: Tensor, like: Tensor) -> torch.nn.Parameter:
    return torch.nn.Parameter(t, requires_grad=like.requires_grad)


def _zero_rows(v: Tensor, n: int) -> Tensor:
    return torch.zeros((n,) + tuple(v.shape[1:]), device=v.device, dtype=v.dtype)


@torch.no_grad()
def duplicate(params: Params, optimizers: Dict[str, torch.optim.Optimizer], state: Dict[str, Tensor], mask: Tensor):
    """Append a copy of every Gaussian selected by ``mask`` (fresh optimizer moments, inherited statistics)."""
    sel = torch.where(mask)[0]
    _update_param_with_optimizer(lambda name, p: _as_param(torch.cat([p, p[sel]]), p),
                                 lambda key, v: torch.cat([v, _zero_rows(v, len(sel))]), params, optimizers)
    for k, v in state.items():
        if isinstance(v, Tensor):
            state[k] = torch.cat([v, v[sel]])


@torch.no_grad()
def split(params: Params, optimizers: Dict[str, torch.optim.Optimizer], state: Dict[str, Tensor], mask: Tensor,
          revised_opacity: bool = False):
    """Replace every selected Gaussian by two samples of itself: means drawn from N(mean, Sigma), scales / 1.6, and
    (``revised_opacity``, arXiv 2404.06109) opacity 1 - sqrt(1 - o). Unselected rows come first in the new order."""
    device = mask.device
    sel, rest = torch.where(mask)[0], torch.where(~mask)[0]
    scales = torch.exp(params["scales"][sel])
    rot = _quat_to_rotmat(F.normalize(params["quats"][sel], dim=-1))
    offsets = torch.einsum("nij,nj,bnj->bni", rot, scales, torch.randn(2, len(sel), 3, device=device))  # [2, n, 3]

    def param_fn(name: str, p: Tensor) -> Tensor:
        twice = [2] + [1] * (p.dim() - 1)
        if name == "means":
            new = (p[sel] + offsets).reshape(-1, 3)
        elif name == "scales":
            new = torch.log(scales / 1.6).repeat(2, 1)
        elif name == "opacities" and revised_opacity:
            new = torch.logit(1.0 - torch.sqrt(1.0 - torch.sigmoid(p[sel]))).repeat(twice)
        else:
            new = p[sel].repeat(twice)
        return _as_param(torch.cat([p[rest], new]), p)

    _update_param_with_optimizer(param_fn, lambda key, v: torch.cat([v[rest], _zero_rows(v, 2 * len(sel))]), params,
                                 optimizers)
    for k, v in state.items():
        if isinstance(v, Tensor):
            state[k] = torch.cat([v[rest], v[sel].repeat([2] + [1] * (v.dim() - 1))])


@torch.no_grad()
def remove(params: Params, optimizers: Dict[str, torch.optim.Optimizer], state: Dict[str, Tensor], mask: Tensor):
    """Drop the Gaussians selected by ``mask``."""
    keep = torch.where(~mask)[0]
    _update_param_with_optimizer(lambda name, p: _as_param(p[keep], p), lambda key, v: v[keep], params, optimizers)
    for k, v in state.items():
        if isinstance(v, Tensor):
            state[k] = v[keep]


@torch.no_grad()
def reset_opa(params: Params, optimizers: Dict[str, torch.optim.Optimizer], state: Dict[str, Tensor], value: float):
    """Clamp the opacities to at most ``value`` (post-sigmoid) and zero their optimizer moments."""
    cap = torch.logit(torch.tensor(value)).item()

    def param_fn(name: str, p: Tensor) -> Tensor:
        assert name == "opacities", name
        return _as_param(torch.clamp(p, max=cap), p)

    _update_param_with_optimizer(param_fn, lambda key, v: torch.zeros_like(v), params, optimizers, names=["opacities"])


def _relocated(params: Params, idx: Tensor, binoms: Tensor, min_opacity: float):
    """Eq. 9 of the MCMC paper for the sampled source rows ``idx`` (a row sampled r times is shared by r + 1 Gaussians)."""
    opacities = torch.sigmoid(params["opacities"])
    return compute_relocation(opacities=opacities[idx], scales=torch.exp(params["scales"])[idx],
                              ratios=torch.bincount(idx)[idx] + 1, binoms=binoms, min_opacity=min_opacity)


@torch.no_grad()
def relocate(params: Params, optimizers: Dict[str, torch.optim.Optimizer], state: Dict[str, Tensor], mask: Tensor,
             binoms: Tensor, min_opacity: float = 0.005):
    """Teleport the dead Gaussians (``mask``) onto live ones sampled in proportion to opacity; source and copies get
    the relocated opacity / scale and fresh optimizer moments."""
    dead, alive = mask.nonzero(as_tuple=True)[0], (~mask).nonzero(as_tuple=True)[0]
    probs = torch.sigmoid(params["opacities"])[alive].flatten()
    src = alive[_multinomial_sample(probs, len(dead), replacement=True)]
    new_opacities, new_scales = _relocated(params, src, binoms, min_opacity)

    def param_fn(name: str, p: Tensor) -> Tensor:
        if name == "opacities":
            p[src] = torch.logit(new_opacities)
        elif name == "scales":
            p[src] = torch.log(new_scales)
        p[dead] = p[src]
        return _as_param(p, p)

    def optimizer_fn(key: str, v: Tensor) -> Tensor:
        v[src] = 0
        return v

    _update_param_with_optimizer(param_fn, optimizer_fn, params, optimizers)
    for k, v in state.items():
        if isinstance(v, Tensor):
            v[src] = 0


@torch.no_grad()
def sample_add(params: Params, optimizers: Dict[str, torch.optim.Optimizer], state: Dict[str, Tensor], n: int,
               binoms: Tensor, min_opacity: float = 0.005):
    """Grow the set by ``n`` Gaussians cloned from rows sampled in proportion to opacity (relocated opacity / scale)."""
    src = _multinomial_sample(torch.sigmoid(params["opacities"]).flatten(), n, replacement=True)
    new_opacities, new_scales = _relocated(params, src, binoms, min_opacity)

    def param_fn(name: str, p: Tensor) -> Tensor:
        if name == "opacities":
            p[src] = torch.logit(new_opacities)
        elif name == "scales":
            p[src] = torch.log(new_scales)
        return _as_param(torch.cat([p, p[src]]), p)

    _update_param_with_optimizer(param_fn, lambda key, v: torch.cat([v, _zero_rows(v, len(src))]), params, optimizers)
    for k, v in state.items():
        if isinstance(v, Tensor):
            state[k] = torch.cat([v, _zero_rows(v, len(src))])


@torch.no_grad()
def inject_noise_to_position(params: Params, optimizers: Dict[str, torch.optim.Optimizer], state: Dict[str, Tensor],
                             scaler: float, t: float = 0.005, k: float = 100.0):
    """SGLD noise of the MCMC strategy: means += Sigma (eps * sigmoid(-k (opacity - t)) * scaler), eps ~ N(0, I) — one
    fused kernel (``gsx_mcmc_perturb``) on the raw parameters (log-scales, opacity logits, un-normalised quats)."""
    means = params["means"]
    noise = torch.randn_like(means)
    torch.ops.gsplat.mcmc_perturb_positions(means.data, params["quats"].data, params["scales"].data,
                                            params["opacities"].data.reshape(-1), noise, float(scaler), float(t), float(k))


# ----------------------------------------------------------------------------------------------------------------------
# Row plans: ONE rewrite of the model per refinement
# ----------------------------------------------------------------------------------------------------------------------
# The reference refines in stages (duplicate, then split, then remove; relocate, then sample_add), and every stage rebuilds
# every parameter, both Adam moments of every optimizer and the running statistics - three full passes over a model of
# millions of rows (59 floats per row with degree-3 SH, x3 with the moments). The strategies of this package instead derive
# a single RowPlan for the whole refinement from a few per-row vectors (opacity, largest scale, statistics) and apply it
# once: every tensor is gathered exactly one time.
class RowPlan:
    """New row j of the Gaussian set is a copy of old row ``src[j]``.

    ``fresh[j]``: its optimizer moments start at zero instead of being inherited. ``values[name]`` = (rows, tensor):
    after the gather, ``new[name][rows] = tensor`` (rows index the NEW set). ``moment_src[j]`` (default ``src``): the old
    row whose optimizer moments new row j inherits - MCMC relocation teleports a dead row onto a live one but leaves the
    dead row's OWN Adam moments in place (reference ``gsplat/strategy/ops.py`` ``relocate``: only ``v[sampled_idxs] = 0``)."""

    def __init__(self, src: Tensor, fresh: Tensor, moment_src: Tensor = None):
        self.src, self.fresh = src, fresh
        self.moment_src = moment_src
        self.values: Dict[str, tuple] = {}

    def set(self, name: str, rows: Tensor, values: Tensor) -> None:
        self.values[name] = (rows, values)

    def select(self, keep: Tensor) -> "RowPlan":
        """The plan restricted to the new rows selected by the bool mask ``keep`` (row overrides are renumbered)."""
        out = RowPlan(self.src[keep], self.fresh[keep], None if self.moment_src is None else self.moment_src[keep])
        new_index = torch.cumsum(keep, 0) - 1
        for name, (rows, vals) in self.values.items():
            k = keep[rows]
            out.values[name] = (new_index[rows[k]], vals[k])
        return out

    def __len__(self) -> int:
        return int(self.src.numel())


@torch.no_grad()
def apply_plan(params: Params, optimizers: Dict[str, torch.optim.Optimizer], state: Dict[str, Tensor], plan: RowPlan,
               zero_state: bool = False) -> None:
    """Rebuild every parameter, its optimizer moments and the per-Gaussian statistics in ``state`` through ``plan``
    (statistics: inherited through the plan, or zeroed when ``zero_state``)."""
    src, fresh = plan.src, plan.fresh
    msrc = src if plan.moment_src is None else plan.moment_src
    n_old = len(next(iter(params.values())))
    any_fresh = bool(fresh.any().item()) if fresh.numel() else False

    def param_fn(name: str, p: Tensor) -> Tensor:
        new = p.detach()[src]
        if name in plan.values:
            rows, vals = plan.values[name]
            new[rows] = vals.to(new.dtype).reshape((rows.numel(),) + tuple(new.shape[1:]))
        return _as_param(new, p)

    def optimizer_fn(key: str, v: Tensor) -> Tensor:
        new = v[msrc]
        if any_fresh:
            new[fresh] = 0
        return new

    _update_param_with_optimizer(param_fn, optimizer_fn, params, optimizers)
    for k, v in state.items():
        if isinstance(v, Tensor) and v.dim() >= 1 and v.shape[0] == n_old:
            state[k] = (torch.zeros((len(plan),) + tuple(v.shape[1:]), device=v.device, dtype=v.dtype) if zero_state
                        else v[src])
