"""3D Gaussian Splatting as Markov Chain Monte Carlo (arXiv 2404.09591) - same class name, fields, defaults and callbacks as
the reference (``gsplat/strategy/mcmc.py:38-239``): dead Gaussians are teleported onto live ones, the set grows by 5 % per
refinement up to ``cap_max``, and the means receive covariance-shaped SGLD noise after every step (one fused kernel,
``gsx_mcmc_perturb``).

A refinement is one ``RowPlan`` (``strategy/ops.py``). The reference relocates first and then samples the additions from
the relocated model - two rewrites of every parameter and optimizer moment; only the opacities and scales take part in
that dependency, so both stages are planned on those two vectors and the model is rewritten once."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Any, Dict, Union

import torch
from torch import Tensor

from ..relocation import compute_relocation
from .base import Strategy
from .ops import RowPlan, _multinomial_sample, apply_plan, inject_noise_to_position

Params = Union[Dict[str, torch.nn.Parameter], torch.nn.ParameterDict]


@dataclass
class MCMCStrategy(Strategy):
    """Call ``step_post_backward(..., lr=<learning rate of the means>)``."""

    cap_max: int = 1_000_000
    noise_lr: float = 5e5
    refine_start_iter: int = 500
    refine_stop_iter: int = 25_000
    noise_injection_stop_iter: int = -1
    refine_every: int = 100
    min_opacity: float = 0.005
    verbose: bool = False
    noise_opacity_t: float = 0.005
    noise_opacity_k: float = 100.0
    # The reference returns the allocator's cached blocks to the driver after every refinement (torch.cuda.empty_cache(),
    # sized for 24 GB cards). That is a hipFree / hipMalloc round trip for the working set in the steps that follow (measured
    # on the 1 M-Gaussian training step: 7.3 instead of 5.4 ms for the refinement step), and 288 GB of HBM has no use for it:
    # opt-in here.
    release_cached_memory: bool = False

    def initialize_state(self) -> Dict[str, Any]:
        top = 51  # binomial coefficients C(n, k) for n < 51 (Eq. 9 of the paper sums over the copies of a Gaussian)
        table = torch.tensor([[float(math.comb(n, k)) if k <= n else 0.0 for k in range(top)] for n in range(top)])
        return {"binoms": table}

    def check_sanity(self, params: Params, optimizers: Dict[str, torch.optim.Optimizer]):
        super().check_sanity(params, optimizers)
        missing = [k for k in ("means", "scales", "quats", "opacities") if k not in params]
        assert not missing, f"params lacks {missing}"

    def step_post_backward(self, params: Params, optimizers, state: Dict[str, Any], step: int, info: Dict[str, Any],
                           lr: float):
        state["binoms"] = state["binoms"].to(params["means"].device)
        if self.refine_start_iter < step < self.refine_stop_iter and step % self.refine_every == 0:
            n_moved, n_new = self._refine(params, optimizers, state["binoms"])
            if self.verbose:
                print(f"step {step}: {n_moved} relocated, {n_new} added -> {len(params['means'])} Gaussians")
            if self.release_cached_memory:
                torch.cuda.empty_cache()
        if self.noise_injection_stop_iter < 0 or step < self.noise_injection_stop_iter:
            inject_noise_to_position(params=params, optimizers=optimizers, state={}, scaler=lr * self.noise_lr,
                                     t=self.noise_opacity_t, k=self.noise_opacity_k)

    @torch.no_grad()
    def _refine(self, params: Params, optimizers, binoms: Tensor):
        dev = params["means"].device
        n = len(params["means"])
        opacity = torch.sigmoid(params["opacities"].flatten()).clone()
        scale = torch.exp(params["scales"]).clone()
        src = torch.arange(n, device=dev)
        msrc = torch.arange(n, device=dev)  # optimizer moments: a teleported (dead) row keeps its own, as in the reference
        fresh = torch.zeros(n, dtype=torch.bool, device=dev)

        def share(rows: Tensor) -> None:
            """``rows`` (with repetitions) are about to be copied: a row sampled r times is shared by r + 1 Gaussians and
            all of them get the opacity / scale of Eq. 9 (written back into the two planning vectors)."""
            o, s = compute_relocation(opacities=opacity[rows], scales=scale[rows], ratios=torch.bincount(rows)[rows] + 1,
                                      binoms=binoms, min_opacity=self.min_opacity)
            opacity[rows], scale[rows] = o, s

        # stage 1: dead rows become copies of live rows drawn in proportion to opacity
        dead = (opacity <= self.min_opacity).nonzero(as_tuple=True)[0]
        touched = torch.zeros(n, dtype=torch.bool, device=dev)
        if len(dead):
            alive = (opacity > self.min_opacity).nonzero(as_tuple=True)[0]
            pick = alive[_multinomial_sample(opacity[alive], len(dead), replacement=True)]
            share(pick)
            src[dead] = pick
            opacity[dead], scale[dead] = opacity[pick], scale[pick]
            fresh[pick] = True  # the reference zeroes the moments of the sources (not of the teleported rows)
            touched[pick] = True
            touched[dead] = True
        # stage 2: 5 % more rows (up to cap_max), drawn from the relocated opacities
        n_new = max(0, min(self.cap_max, int(1.05 * n)) - n)
        if n_new:
            pick2 = _multinomial_sample(opacity, n_new, replacement=True)
            share(pick2)
            touched[pick2] = True
            rows_src = torch.cat([src, src[pick2]])
            plan = RowPlan(rows_src, torch.cat([fresh, torch.ones(n_new, dtype=torch.bool, device=dev)]),
                           torch.cat([msrc, msrc[pick2]]))
            opacity, scale = torch.cat([opacity, opacity[pick2]]), torch.cat([scale, scale[pick2]])
            touched = torch.cat([touched, torch.ones(n_new, dtype=torch.bool, device=dev)])
        else:
            plan = RowPlan(src, fresh, msrc)
        rows = touched.nonzero(as_tuple=True)[0]
        if len(rows):
            plan.set("opacities", rows, torch.logit(opacity[rows]).reshape((len(rows),) + tuple(params["opacities"].shape[1:])))
            plan.set("scales", rows, torch.log(scale[rows]))
            apply_plan(params, optimizers, {}, plan)
        return len(dead), n_new
