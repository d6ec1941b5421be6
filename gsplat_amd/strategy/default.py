"""The densification schedule of the original 3DGS paper, optionally with AbsGS gradients - same class name, fields,
defaults and callbacks as the reference (``gsplat/strategy/default.py:30-377``), so a training script switches by changing
an import. Consumes the ``meta`` dict of ``gsplat_amd.rasterization`` (``means2d`` with ``.grad`` / ``.absgrad``, ``radii``,
``gaussian_ids``, ``width``, ``height``, ``n_cameras``).

What differs is HOW a refinement is carried out. The reference edits the model three times in a row (duplicate, split,
prune), each time rebuilding every parameter and every optimizer moment. Here a refinement is PLANNED on four per-row
vectors - mean screen-space gradient, largest scale, opacity, largest screen radius - and executed as one ``RowPlan``
(``strategy/ops.py``): which old row every surviving new row copies, which rows start with fresh optimizer moments, and the
few values that are not copies (sampled positions and shrunken scales of split children, revised opacities). The pruning
test of the reference runs AFTER growth, on the children too; since a child's opacity and scale are known functions of its
parent's, the same test is evaluated on the plan before anything is materialised."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Tuple, Union

import torch
import torch.nn.functional as F
from torch import Tensor

import os

_FUSED_ACCUMULATE = os.environ.get("GSPLAT_AMD_STRATEGY_FUSED", "1") not in ("0", "")  # A/B: 0 = the tensor-op form

from .base import Strategy
from .ops import RowPlan, _quat_to_rotmat, apply_plan, reset_opa

Params = Union[Dict[str, torch.nn.Parameter], torch.nn.ParameterDict]

_SPLIT_SHRINK = 1.6  # scale of a split child = parent's / 1.6 (reference ops.py: split)


@dataclass
class DefaultStrategy(Strategy):
    """>>> strategy = DefaultStrategy(); strategy.check_sanity(params, optimizers); state = strategy.initialize_state()
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
    # The reference returns the allocator's cached blocks to the driver after every refinement (torch.cuda.empty_cache(),
    # sized for 24 GB cards). That is a hipFree / hipMalloc round trip for the working set in the steps that follow (measured
    # on the 1 M-Gaussian training step: 7.3 instead of 5.4 ms for the refinement step), and 288 GB of HBM has no use for it:
    # opt-in here.
    release_cached_memory: bool = False

    # ---- interface -----------------------------------------------------------------------------------------------------
    def initialize_state(self, scene_scale: float = 1.0) -> Dict[str, Any]:
        state: Dict[str, Any] = {"grad2d": None, "count": None, "scene_scale": scene_scale}
        if self.refine_scale2d_stop_iter > 0:
            state["radii"] = None
        return state

    def check_sanity(self, params: Params, optimizers: Dict[str, torch.optim.Optimizer]):
        super().check_sanity(params, optimizers)
        missing = [k for k in ("means", "scales", "quats", "opacities") if k not in params]
        assert not missing, f"params lacks {missing}"

    def step_pre_backward(self, params: Params, optimizers, state: Dict[str, Any], step: int, info: Dict[str, Any]):
        assert self.key_for_gradient in info, f"info['{self.key_for_gradient}'] (the projected means) is needed"
        info[self.key_for_gradient].retain_grad()

    def step_post_backward(self, params: Params, optimizers, state: Dict[str, Any], step: int, info: Dict[str, Any],
                           packed: bool = False):
        if step >= self.refine_stop_iter:
            return
        self._accumulate(params, state, info, packed)
        due = step > self.refine_start_iter and step % self.refine_every == 0
        if due and step % self.reset_every >= self.pause_refine_after_reset:
            counts = self._refine(params, optimizers, state, step)
            if self.verbose:
                print("step {}: +{} duplicated, +{} split, -{} pruned -> {} Gaussians".format(step, *counts,
                                                                                              len(params["means"])))
            if self.release_cached_memory:
                torch.cuda.empty_cache()
        if step > 0 and step % self.reset_every == 0:
            reset_opa(params=params, optimizers=optimizers, state=state, value=self.prune_opa * 2.0)

    # ---- statistics ----------------------------------------------------------------------------------------------------
    def _accumulate(self, params: Params, state: Dict[str, Any], info: Dict[str, Any], packed: bool) -> None:
        """Per Gaussian: summed norm of the screen-space gradient (in [-1, 1] image units, 1/C of a mean loss undone),
        number of views it was visible in, largest relative screen radius."""
        need = ("width", "height", "n_cameras", "radii", "gaussian_ids", self.key_for_gradient)
        lacking = [k for k in need if k not in info]
        assert not lacking, f"info lacks {lacking}"
        g = info[self.key_for_gradient]
        grad = g.absgrad if self.absgrad else g.grad
        half = grad.new_tensor([0.5 * info["width"], 0.5 * info["height"]]) * info["n_cameras"]
        n, dev = len(params["means"]), grad.device
        for key in ("grad2d", "count") + (("radii",) if self.refine_scale2d_stop_iter > 0 else ()):
            if state[key] is None:
                state[key] = torch.zeros(n, device=dev)
        if packed:
            rows, norms, radius = info["gaussian_ids"], (grad * half).norm(dim=-1), info["radii"].amax(dim=-1)
            state["grad2d"].index_add_(0, rows, norms)
            state["count"].index_add_(0, rows, torch.ones_like(norms))
            if self.refine_scale2d_stop_iter > 0:
                rel = radius.to(state["radii"].dtype) / float(max(info["width"], info["height"]))
                state["radii"].scatter_reduce_(0, rows, rel, reduce="amax", include_self=True)
            return
        # Dense rows [..., C, N]: the same sums as masked reductions over the camera axes. The reference gathers the visible
        # pairs first (`torch.where` + three boolean-mask indexings: each a nonzero with a device-to-host read of its size,
        # 0.15 ms of kernels and three pipeline drains per training step at 1 M Gaussians); nothing here leaves the device.
        if self._accumulate_fused(state, info, grad, n):
            return
        seen = (info["radii"] > 0).all(dim=-1).reshape(-1, n)  # [C, N]
        # selects, not products: a non-finite gradient in a row that is NOT visible (the reference's `grads[sel]` never reads
        # such rows) must not turn into NaN * 0 = NaN
        norms = (grad * half).norm(dim=-1).reshape(-1, n)
        state["grad2d"] += torch.where(seen, norms, norms.new_zeros(())).sum(dim=0)
        state["count"] += seen.sum(dim=0).to(state["count"].dtype)
        if self.refine_scale2d_stop_iter > 0:
            rel = info["radii"].amax(dim=-1).reshape(-1, n).to(state["radii"].dtype) / float(max(info["width"], info["height"]))
            state["radii"] = torch.maximum(state["radii"], torch.where(seen, rel, rel.new_zeros(())).amax(dim=0))

    def _accumulate_fused(self, state: Dict[str, Any], info: Dict[str, Any], grad: Tensor, n: int) -> bool:
        """The dense-row sums above as ONE launch (C-ABI gsx_strategy_accumulate, csrc/optim.hip) when everything lives on the GPU
        in float32 / int32: the tensor-op form is ~10 launches (70 us per training step at 1 M Gaussians). The gradient is read in
        place - the retained gradient of means2d is a column view of the compositing backward's gradient rows."""
        if not _FUSED_ACCUMULATE:
            return False
        radii = info["radii"]
        track = self.refine_scale2d_stop_iter > 0
        if not (grad.is_cuda and grad.dtype == torch.float32 and radii.dtype == torch.int32 and radii.is_contiguous()
                and n > 0 and radii.numel() == grad.numel() and grad.numel() % (2 * n) == 0 and grad.shape[-1] == 2
                and all(state[k].dtype == torch.float32 and state[k].is_contiguous() and state[k].is_cuda
                        for k in ("grad2d", "count") + (("radii",) if track else ()))):
            return False
        rows = grad.reshape(-1, 2) if grad.is_contiguous() else None
        stride = 2
        if rows is None:  # [.., N, 2] view with one row stride throughout (a column pair of an array-of-structures buffer)
            st, shp = grad.stride(), grad.shape
            if st[-1] != 1:
                return False
            stride, expect = st[-2], st[-2]
            for d in range(grad.dim() - 2, -1, -1):
                if shp[d] != 1 and st[d] != expect:
                    return False
                expect *= shp[d]
        from .._cabi import call, ptr, ptr_strided

        C = grad.numel() // (2 * n)
        call("gsx_strategy_accumulate", ptr_strided(grad) if rows is None else ptr(rows), int(stride), ptr(radii), C, n,
             0.5 * info["width"] * info["n_cameras"], 0.5 * info["height"] * info["n_cameras"],
             1.0 / float(max(info["width"], info["height"])), ptr(state["grad2d"]), ptr(state["count"]),
             ptr(state["radii"]) if track else None)
        return True

    # ---- one refinement = one plan -------------------------------------------------------------------------------------
    @torch.no_grad()
    def _refine(self, params: Params, optimizers, state: Dict[str, Any], step: int) -> Tuple[int, int, int]:
        dev = params["means"].device
        n = len(params["means"])
        scale = torch.exp(params["scales"])
        largest = scale.amax(dim=-1)
        opacity = torch.sigmoid(params["opacities"].flatten())
        track_radii = self.refine_scale2d_stop_iter > 0
        screen = state["radii"] if track_radii else None

        hot = state["grad2d"] / state["count"].clamp_min(1) > self.grow_grad2d
        small = largest <= self.grow_scale3d * state["scene_scale"]
        clone = hot & small
        split = hot & ~small
        if step < self.refine_scale2d_stop_iter:
            split = split | (screen > self.grow_scale2d)
        stay_rows = (~split).nonzero(as_tuple=True)[0]
        clone_rows = clone.nonzero(as_tuple=True)[0]
        split_rows = split.nonzero(as_tuple=True)[0]
        n_stay, n_clone, n_split = len(stay_rows), len(clone_rows), len(split_rows)

        # grown set = originals that are not split | clones | two children per split row
        src = torch.cat([stay_rows, clone_rows, split_rows, split_rows])
        fresh = torch.cat([torch.zeros(n_stay, dtype=torch.bool, device=dev),
                           torch.ones(n_clone + 2 * n_split, dtype=torch.bool, device=dev)])
        plan = RowPlan(src, fresh)
        child_rows = torch.arange(n_stay + n_clone, len(src), device=dev)
        new_opacity, new_largest = opacity[src], largest[src]
        if n_split:
            rot = _quat_to_rotmat(F.normalize(params["quats"][split_rows], dim=-1))
            jitter = torch.einsum("nij,nj,bnj->bni", rot, scale[split_rows], torch.randn(2, n_split, 3, device=dev))
            plan.set("means", child_rows, (params["means"][split_rows] + jitter).reshape(-1, 3))
            plan.set("scales", child_rows, torch.log(scale[split_rows] / _SPLIT_SHRINK).repeat(2, 1))
            new_largest[child_rows] = new_largest[child_rows] / _SPLIT_SHRINK
            if self.revised_opacity:  # arXiv 2404.06109
                revised = 1.0 - torch.sqrt(1.0 - opacity[split_rows])
                plan.set("opacities", child_rows, torch.logit(revised).repeat(2).reshape(
                    (2 * n_split,) + tuple(params["opacities"].shape[1:])))
                new_opacity[child_rows] = revised.repeat(2)

        # the reference prunes AFTER growing: the same test, evaluated on the planned rows
        doomed = new_opacity < self.prune_opa
        if step > self.reset_every:
            doomed = doomed | (new_largest > self.prune_scale3d * state["scene_scale"])
            if step < self.refine_scale2d_stop_iter:
                doomed = doomed | (screen[src] > self.prune_scale2d)
        n_prune = int(doomed.sum().item())
        if n_prune:
            plan = plan.select(~doomed)
        if n_clone or n_split or n_prune:
            apply_plan(params, optimizers, state, plan, zero_state=True)
        else:
            for key in ("grad2d", "count") + (("radii",) if track_radii else ()):
                state[key].zero_()
        return n_clone, n_split, n_prune
