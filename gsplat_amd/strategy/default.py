"""The densification schedule of the original 3DGS paper, optionally with AbsGS gradients (reference
``gsplat/strategy/default.py:30-377``): duplicate small Gaussians and split large ones whose screen-space gradient is
high, prune transparent / oversized ones, reset opacities periodically. Consumes the ``meta`` dict returned by
``gsplat_amd.rasterization`` (``means2d`` with its ``.grad`` / ``.absgrad``, ``radii``, ``gaussian_ids``, ``width``,
``height``, ``n_cameras``)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Tuple, Union

import torch

from .base import Strategy
from .ops import duplicate, remove, reset_opa, split

Params = Union[Dict[str, torch.nn.Parameter], torch.nn.ParameterDict]


@dataclass
class DefaultStrategy(Strategy):
    """Same fields and defaults as the reference class.

    >>> strategy = DefaultStrategy(); strategy.check_sanity(params, optimizers); state = strategy.initialize_state()
    >>> for step in range(n):
    ...     colors, alphas, info = rasterization(...)
    ...     strategy.step_pre_backward(params, optimizers, state, step, info)
    ...     loss.backward()
    ...     strategy.step_post_backward(params, optimizers, state, step, info)
    """

    prune_opa: float = 0.005
    grow_grad2d: float = 0.0002
    grow_scale3d: float = 0.01
    grow_scale2d: float = 0.05
    prune_scale3d: float = 0.1
    prune_scale2d: float = 0.15
    refine_scale2d_stop_iter: int = 0
    refine_start_iter: int = 500
    refine_stop_iter: int = 15_000
    reset_every: int = 3000
    refine_every: int = 100
    pause_refine_after_reset: int = 0
    absgrad: bool = False
    revised_opacity: bool = False
    verbose: bool = False
    key_for_gradient: str = "means2d"  # "gradient_2dgs" for rasterization_2dgs

    def initialize_state(self, scene_scale: float = 1.0) -> Dict[str, Any]:
        state: Dict[str, Any] = {"grad2d": None, "count": None, "scene_scale": scene_scale}
        if self.refine_scale2d_stop_iter > 0:
            state["radii"] = None
        return state

    def check_sanity(self, params: Params, optimizers: Dict[str, torch.optim.Optimizer]):
        super().check_sanity(params, optimizers)
        for key in ("means", "scales", "quats", "opacities"):
            assert key in params, f"{key} is required in params but missing."

    def step_pre_backward(self, params: Params, optimizers, state: Dict[str, Any], step: int, info: Dict[str, Any]):
        assert self.key_for_gradient in info, "The 2D means of the Gaussians is required but missing."
        info[self.key_for_gradient].retain_grad()

    def step_post_backward(self, params: Params, optimizers, state: Dict[str, Any], step: int, info: Dict[str, Any],
                           packed: bool = False):
        if step >= self.refine_stop_iter:
            return
        self._update_state(params, state, info, packed=packed)
        refine_now = (step > self.refine_start_iter and step % self.refine_every == 0
                      and step % self.reset_every >= self.pause_refine_after_reset)
        if refine_now:
            n_dupli, n_split = self._grow_gs(params, optimizers, state, step)
            if self.verbose:
                print(f"Step {step}: {n_dupli} GSs duplicated, {n_split} GSs split. Now having {len(params['means'])} GSs.")
            n_prune = self._prune_gs(params, optimizers, state, step)
            if self.verbose:
                print(f"Step {step}: {n_prune} GSs pruned. Now having {len(params['means'])} GSs.")
            state["grad2d"].zero_()
            state["count"].zero_()
            if self.refine_scale2d_stop_iter > 0:
                state["radii"].zero_()
            torch.cuda.empty_cache()
        if step % self.reset_every == 0 and step > 0:
            reset_opa(params=params, optimizers=optimizers, state=state, value=self.prune_opa * 2.0)

    # -- running statistics: summed screen-space gradient norm and visibility count per Gaussian ---------------------
    def _update_state(self, params: Params, state: Dict[str, Any], info: Dict[str, Any], packed: bool = False):
        for key in ("width", "height", "n_cameras", "radii", "gaussian_ids", self.key_for_gradient):
            assert key in info, f"{key} is required but missing."
        g = info[self.key_for_gradient]
        grads = (g.absgrad if self.absgrad else g.grad).clone()
        # gradients are w.r.t. pixel coordinates: normalise to [-1, 1] image coordinates, undo the 1/C of a mean loss
        grads[..., 0] *= info["width"] / 2.0 * info["n_cameras"]
        grads[..., 1] *= info["height"] / 2.0 * info["n_cameras"]
        n = len(next(iter(params.values())))
        dev = grads.device
        if state["grad2d"] is None:
            state["grad2d"] = torch.zeros(n, device=dev)
        if state["count"] is None:
            state["count"] = torch.zeros(n, device=dev)
        if self.refine_scale2d_stop_iter > 0 and state["radii"] is None:
            state["radii"] = torch.zeros(n, device=dev)
        if packed:
            ids = info["gaussian_ids"]
            radii = info["radii"].max(dim=-1).values
        else:
            visible = (info["radii"] > 0.0).all(dim=-1)  # [C, N]
            ids = torch.where(visible)[1]
            grads = grads[visible]
            radii = info["radii"][visible].max(dim=-1).values
        state["grad2d"].index_add_(0, ids, grads.norm(dim=-1))
        state["count"].index_add_(0, ids, torch.ones_like(ids, dtype=torch.float32))
        if self.refine_scale2d_stop_iter > 0:
            state["radii"][ids] = torch.maximum(state["radii"][ids], radii / float(max(info["width"], info["height"])))

    @torch.no_grad()
    def _grow_gs(self, params: Params, optimizers, state: Dict[str, Any], step: int) -> Tuple[int, int]:
        mean_grad = state["grad2d"] / state["count"].clamp_min(1)
        high = mean_grad > self.grow_grad2d
        small = torch.exp(params["scales"]).max(dim=-1).values <= self.grow_scale3d * state["scene_scale"]
        is_dupli = high & small
        is_split = high & ~small
        if step < self.refine_scale2d_stop_iter:
            is_split |= state["radii"] > self.grow_scale2d
        n_dupli, n_split = int(is_dupli.sum().item()), int(is_split.sum().item())
        if n_dupli > 0:
            duplicate(params=params, optimizers=optimizers, state=state, mask=is_dupli)
        # the duplicates were appended at the end: they are never split in the same round
        is_split = torch.cat([is_split, torch.zeros(n_dupli, dtype=torch.bool, device=is_split.device)])
        if n_split > 0:
            split(params=params, optimizers=optimizers, state=state, mask=is_split, revised_opacity=self.revised_opacity)
        return n_dupli, n_split

    @torch.no_grad()
    def _prune_gs(self, params: Params, optimizers, state: Dict[str, Any], step: int) -> int:
        is_prune = torch.sigmoid(params["opacities"].flatten()) < self.prune_opa
        if step > self.reset_every:
            too_big = torch.exp(params["scales"]).max(dim=-1).values > self.prune_scale3d * state["scene_scale"]
            if step < self.refine_scale2d_stop_iter:
                too_big |= state["radii"] > self.prune_scale2d
            is_prune |= too_big
        n_prune = int(is_prune.sum().item())
        if n_prune > 0:
            remove(params=params, optimizers=optimizers, state=state, mask=is_prune)
        return n_prune
