"""3D Gaussian Splatting as Markov Chain Monte Carlo (arXiv 2404.09591; reference ``gsplat/strategy/mcmc.py:38-239``):
dead Gaussians are teleported onto live ones, the set grows by 5 % per refinement up to ``cap_max``, and the means
receive covariance-shaped SGLD noise after every step."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Any, Dict, Union

import torch
from torch import Tensor

from .base import Strategy
from .ops import inject_noise_to_position, relocate, sample_add

Params = Union[Dict[str, torch.nn.Parameter], torch.nn.ParameterDict]


@dataclass
class MCMCStrategy(Strategy):
    """Same fields and defaults as the reference class; call ``step_post_backward(..., lr=<means learning rate>)``."""

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

    def initialize_state(self) -> Dict[str, Any]:
        n_max = 51  # table of binomial coefficients C(n, k), n < 51
        binoms = torch.zeros((n_max, n_max))
        for n in range(n_max):
            for k in range(n + 1):
                binoms[n, k] = math.comb(n, k)
        return {"binoms": binoms}

    def check_sanity(self, params: Params, optimizers: Dict[str, torch.optim.Optimizer]):
        super().check_sanity(params, optimizers)
        for key in ("means", "scales", "quats", "opacities"):
            assert key in params, f"{key} is required in params but missing."

    def step_post_backward(self, params: Params, optimizers, state: Dict[str, Any], step: int, info: Dict[str, Any],
                           lr: float):
        state["binoms"] = state["binoms"].to(params["means"].device)
        binoms = state["binoms"]
        if self.refine_start_iter < step < self.refine_stop_iter and step % self.refine_every == 0:
            n_moved = self._relocate_gs(params, optimizers, binoms)
            if self.verbose:
                print(f"Step {step}: Relocated {n_moved} GSs.")
            n_new = self._add_new_gs(params, optimizers, binoms)
            if self.verbose:
                print(f"Step {step}: Added {n_new} GSs. Now having {len(params['means'])} GSs.")
            torch.cuda.empty_cache()
        stop = self.noise_injection_stop_iter if self.noise_injection_stop_iter >= 0 else float("inf")
        if step < stop:
            inject_noise_to_position(params=params, optimizers=optimizers, state={}, scaler=lr * self.noise_lr,
                                     t=self.noise_opacity_t, k=self.noise_opacity_k)

    @torch.no_grad()
    def _relocate_gs(self, params: Params, optimizers, binoms: Tensor) -> int:
        dead = torch.sigmoid(params["opacities"].flatten()) <= self.min_opacity
        n = int(dead.sum().item())
        if n > 0:
            relocate(params=params, optimizers=optimizers, state={}, mask=dead, binoms=binoms,
                     min_opacity=self.min_opacity)
        return n

    @torch.no_grad()
    def _add_new_gs(self, params: Params, optimizers, binoms: Tensor) -> int:
        current = len(params["means"])
        n = max(0, min(self.cap_max, int(1.05 * current)) - current)
        if n > 0:
            sample_add(params=params, optimizers=optimizers, state={}, n=n, binoms=binoms, min_opacity=self.min_opacity)
        return n
