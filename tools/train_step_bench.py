#!/usr/bin/env python
"""One TRAINING step around the rasterizer on the MI355X, timed end to end: what examples/simple_trainer.py does per
iteration (reference examples/simple_trainer.py:795-1170: rasterization :722, loss = lerp(L1, 1 - SSIM, 0.2) :951-961, backward, the optimizers' step
:1137-1150, strategy.step_post_backward :1156-1166), on the c3-sized model - 1 M Gaussians, one 1080p camera, SH degree 3,
parameters in the trainer's layout (log-scales, logit-opacities, sh0 / shN stored apart and optimised with their own
rates), SelectiveAdam (visibility-masked fused Adam) and DefaultStrategy (densification statistics every step, ONE
refinement inside the timed window).

Prints one JSON object; `bench.py` embeds it as the "train_step" sub-record of the N = 1 line.
usage: train_step_bench.py [--steps 100] [--gaussians 1000000] [--packed]"""
import argparse
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(steps=100, n_gaussians=1_000_000, packed=False, device=None, refine_every=None, grow_grad2d=2e-7,
        release_cached_memory=False, refine=True, split_sh=False, ssim_lambda=0.2):
    import bench
    import gsplat_amd
    from gsplat_amd.losses import ssim_loss

    dev = device or torch.device("cuda", 0)
    sc, W, H = bench.make_workload(n_gaussians, dev)
    with torch.no_grad():
        target, _, _ = gsplat_amd.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["colors"],
                                                sc["viewmats"], sc["Ks"], W, H, sh_degree=3, packed=packed)
        target = target.clamp(0, 1)
    g = torch.Generator().manual_seed(1)
    noise = lambda t, s: (t + s * torch.randn(t.shape, generator=g).to(dev))  # noqa: E731
    params = torch.nn.ParameterDict({
        "means": torch.nn.Parameter(noise(sc["means"], 5e-3)),
        "quats": torch.nn.Parameter(sc["quats"].clone()),
        "scales": torch.nn.Parameter(torch.log(sc["scales"])),
        "opacities": torch.nn.Parameter(torch.logit(sc["opacities"].clamp(1e-3, 1 - 1e-3))),
        "sh0": torch.nn.Parameter(noise(sc["colors"][:, :1, :], 0.05)),
        "shN": torch.nn.Parameter(sc["colors"][:, 1:, :].clone()),
    })
    lrs = dict(means=1.6e-4, quats=1e-3, scales=5e-3, opacities=5e-2, sh0=2.5e-3, shN=2.5e-3 / 20)  # simple_trainer.py:232-241
    opts = {k: gsplat_amd.SelectiveAdam([{"params": params[k], "lr": lrs[k], "name": k}], eps=1e-15, betas=(0.9, 0.999))
            for k in params.keys()}
    # Refinements at steps R, 2 R, ...: the first one falls into the warm-up (first-use costs of the refinement's own
    # kernels and allocations), the second one into the timed window - steps R + 1 .. R + steps, R = steps / 2.
    # Thresholds low enough that a refinement really edits the model (the fit starts next to its optimum, so the
    # screen-space gradients are far below the trainer's default 2e-4).
    R = max(steps // 2 if refine_every is None else refine_every, 1)
    refine_at = 2 * R
    if not refine:  # profiling runs: plain steps only (three warm-up steps, no refinement anywhere)
        R, refine_at = 2, 10**9
    strategy = gsplat_amd.DefaultStrategy(refine_start_iter=(R - 1) if refine else 10**9, refine_every=R, reset_every=10**9,
                                          refine_stop_iter=refine_at + 1, grow_grad2d=grow_grad2d, verbose=False,
                                          release_cached_memory=release_cached_memory)
    strategy.check_sanity(params, opts)
    state = strategy.initialize_state(scene_scale=1.0)

    def step(i):
        # split_sh: (sh0, shN) go to the split SH kernels as they are stored; otherwise the reference trainer's concatenation
        colors = (params["sh0"], params["shN"]) if split_sh else torch.cat([params["sh0"], params["shN"]], 1)
        rc, ra, info = gsplat_amd.rasterization(params["means"], params["quats"], torch.exp(params["scales"]),
                                                torch.sigmoid(params["opacities"]), colors, sc["viewmats"], sc["Ks"], W, H,
                                                sh_degree=3, packed=packed)
        # the reference trainer's loss (examples/simple_trainer.py:951-961): lerp(L1, 1 - SSIM, ssim_lambda = 0.2)
        l1 = (rc - target).abs().mean()
        if ssim_lambda > 0.0:
            loss = torch.lerp(l1, ssim_loss(rc.permute(0, 3, 1, 2), target.permute(0, 3, 1, 2)), ssim_lambda)
        else:
            loss = l1
        strategy.step_pre_backward(params, opts, state, i, info)
        loss.backward()
        if packed:
            vis = torch.zeros(len(params["means"]), dtype=torch.bool, device=dev).index_fill_(0, info["gaussian_ids"], True)
        else:
            vis = (info["radii"] > 0).all(-1).any(0)
        for o in opts.values():
            o.step(vis)
            o.zero_grad(set_to_none=True)
        strategy.step_post_backward(params, opts, state, i, info, packed=packed)
        return loss

    n_start = len(params["means"])
    for i in range(R + 1):  # warm-up outside the window: allocator, lazy optimizer state, and the FIRST refinement (step R)
        step(i)
    torch.cuda.synchronize()
    n0 = len(params["means"])
    per_step = []
    t0 = time.perf_counter()
    for i in range(R + 1, R + 1 + steps):
        ts = time.perf_counter()
        loss = step(i)
        if i == refine_at or i == refine_at - 1:
            torch.cuda.synchronize()
            per_step.append((i, time.perf_counter() - ts))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    n1 = len(params["means"])
    refine_ms = [round(t * 1e3, 3) for (i, t) in per_step if i == refine_at]
    return {
        "workload": f"training step on the c3-sized model: {n0} Gaussians, 1x{W}x{H}, SH deg 3, sh0 / shN apart, "
                    f"packed={packed}; rasterization + lerp(L1, 1 - SSIM, {ssim_lambda}) + backward + SelectiveAdam (6 tensors) + "
                    "DefaultStrategy",
        "ssim_lambda": ssim_lambda,
        "steps": steps, "ms_per_step": round(wall / steps * 1e3, 4), "steps_per_s": round(steps / wall, 2),
        "mpixels_per_s": round(W * H * steps / wall / 1e6, 2),
        "refinement_at_step": refine_at, "refinement_step_ms": refine_ms[0] if refine_ms else None,
        "gaussians_at_start": n_start, "gaussians_before": n0, "gaussians_after": n1,
        "release_cached_memory": release_cached_memory, "split_sh": split_sh, "final_loss": round(float(loss.detach()), 6),
        "reference": "examples/simple_trainer.py:795-1170 (rasterization :722, optimizer + strategy steps :1137-1166)",
    }


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--packed", action="store_true")
    ap.add_argument("--release-cached-memory", action="store_true", help="torch.cuda.empty_cache() after a refinement, as the reference does")
    ap.add_argument("--split-sh", action="store_true", help="pass colors=(sh0, shN) instead of torch.cat([sh0, shN], 1)")
    ap.add_argument("--ssim-lambda", type=float, default=0.2, help="weight of the SSIM term (the reference's default 0.2; 0 = L1 only)")
    a = ap.parse_args()
    print(json.dumps(run(a.steps, a.gaussians, a.packed, release_cached_memory=a.release_cached_memory, split_sh=a.split_sh,
                         ssim_lambda=a.ssim_lambda)))
