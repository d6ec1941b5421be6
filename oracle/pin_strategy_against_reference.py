"""Pins gsplat_amd/strategy/default.py against the REFERENCE's gsplat/strategy (TEST INFRASTRUCTURE; needs /root/reference).

The reference's DefaultStrategy and its ops (duplicate / split / remove / reset_opa, gsplat/strategy/ops.py:141-300; statistics,
grow and prune rules, gsplat/strategy/default.py:172-390) run on CPU tensors. This script drives BOTH implementations through the
same seeded sequence of step_post_backward() calls - statistics every step, two refinements, one opacity reset, dense and packed
`info` - and compares after every step: the number of Gaussians, every parameter, both Adam moments of every optimizer and the
strategy state. Splits draw `torch.randn(2, n_split, 3)` in both, so on the CPU the children agree bit for bit under one seed.
It then writes tests/golden/strategy_ref.npz (the reference's trajectory) for tests/test_strategy_plan.py, which replays this
repository's side without a reference checkout (CPU exactly; GPU: everything that does not depend on the device's RNG stream).
The MCMC strategy's relocation calls the reference's CUDA op (gsplat/relocation.py) and cannot run here: not pinned, said so.
usage: python oracle/pin_strategy_against_reference.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.environ.get("GSPLAT_REFERENCE_PATH", "/root/reference"))

N0, W, H = 600, 64, 48
STEPS = 7  # refinements at steps 2, 4 and 6 (refine_every = 2, refine_start_iter = 1), opacity reset at step 6 (reset_every = 6)
CFG = dict(prune_opa=0.02, grow_grad2d=0.4, grow_scale3d=0.05, grow_scale2d=0.3, prune_scale3d=0.5, prune_scale2d=0.6,
           refine_scale2d_stop_iter=100, refine_start_iter=1, refine_stop_iter=1000, reset_every=6, refine_every=2,
           revised_opacity=False, verbose=False)


def make_model(seed=0):
    g = torch.Generator().manual_seed(seed)
    p = {
        "means": torch.randn(N0, 3, generator=g),
        "quats": torch.randn(N0, 4, generator=g),
        "scales": torch.log(torch.rand(N0, 3, generator=g) * 0.12 + 0.005),
        "opacities": torch.logit(torch.rand(N0, generator=g) * 0.9 + 0.005),
        "sh0": torch.rand(N0, 1, 3, generator=g),
        "shN": torch.randn(N0, 3, 3, generator=g) * 0.1,
    }
    params = torch.nn.ParameterDict({k: torch.nn.Parameter(v.clone()) for k, v in p.items()})
    opts = {k: torch.optim.Adam([{"params": params[k], "lr": 1e-3, "name": k}]) for k in params}
    for k, o in opts.items():  # populated Adam state (what a refinement has to carry / reset)
        params[k].grad = torch.randn(params[k].shape, generator=g) * 0.01
        o.step()
        o.zero_grad(set_to_none=True)
    return params, opts


def make_info(n, step, packed, C, seed=1):
    """What rasterization() hands to the strategy: radii, the projected means with their gradient, ids when packed."""
    g = torch.Generator().manual_seed(seed * 1000 + step)
    radii = (torch.rand(C, n, 2, generator=g) * 30).to(torch.int32)
    radii[torch.rand(C, n, generator=g) < 0.3] = 0  # invisible pairs
    grad = torch.randn(C, n, 2, generator=g) * 0.01
    grad[(radii <= 0).any(-1)] = float("nan") if step == 3 else 0.0  # garbage in invisible rows must not be read (dense)
    info = {"width": W, "height": H, "n_cameras": C}
    if packed:
        vis = (radii > 0).all(-1)
        cam, gid = torch.where(vis)
        m2 = torch.zeros(len(gid), 2, requires_grad=True)
        m2.grad = grad[vis].clone()
        info.update(radii=radii[vis], gaussian_ids=gid, camera_ids=cam, means2d=m2)
    else:
        m2 = torch.zeros(C, n, 2, requires_grad=True)
        m2.grad = grad.clone()
        info.update(radii=radii, gaussian_ids=None, means2d=m2)
    return info


def snapshot(params, opts, state):
    out = {f"p_{k}": v.detach().clone() for k, v in params.items()}
    for k, o in opts.items():
        st = o.state[params[k]]
        out[f"m_{k}"], out[f"v_{k}"] = st["exp_avg"].clone(), st["exp_avg_sq"].clone()
    for k in ("grad2d", "count", "radii"):
        if state.get(k) is not None:
            out[f"s_{k}"] = state[k].clone()
    return out


def run(strategy_cls, packed, C, seed=5):
    """C = 2 cameras: screen-size rules off. The reference keeps the screen radius with `state[ids] = maximum(state[ids], r)`,
    which with a Gaussian seen by two cameras keeps ONE of the two values ("should be ideally using scatter max",
    default.py:277) - this repository takes the maximum, so the two only agree when every Gaussian has one row: C = 1."""
    params, opts = make_model()
    strat = strategy_cls(**dict(CFG, refine_scale2d_stop_iter=100 if C == 1 else 0))
    strat.check_sanity(params, opts)
    state = strat.initialize_state(scene_scale=1.0)
    torch.manual_seed(seed)  # the splits' torch.randn
    traj = []
    for step in range(STEPS):
        info = make_info(len(params["means"]), step, packed, C)
        strat.step_post_backward(params, opts, state, step, info, packed=packed)
        traj.append(snapshot(params, opts, state))
    return traj


def main():
    from gsplat.strategy import DefaultStrategy as RefStrategy  # the reference

    from gsplat_amd.strategy import DefaultStrategy as OwnStrategy

    out = {}
    for packed, C in ((False, 2), (True, 2), (False, 1), (True, 1)):
        ref, own = run(RefStrategy, packed, C), run(OwnStrategy, packed, C)
        sizes = [len(s["p_means"]) for s in ref]
        assert len(set(sizes)) >= 3, f"the refinements did not edit the model: {sizes}"
        for step, (a, b) in enumerate(zip(ref, own)):
            assert a.keys() == b.keys(), (step, sorted(a.keys() ^ b.keys()))
            for k in a:
                assert a[k].shape == b[k].shape, (packed, step, k, tuple(a[k].shape), tuple(b[k].shape))
                assert torch.allclose(a[k], b[k], rtol=1e-6, atol=1e-7, equal_nan=True), (packed, step, k,
                                                                                            float((a[k] - b[k]).abs().max()))
            tag = f"{'packed' if packed else 'dense'}{C}"
            out[f"{tag}_sizes"] = np.asarray(sizes)
            if step == STEPS - 1 and (packed, C) in ((False, 2), (True, 1)):  # the fixture keeps two final states + all sizes
                for k, v in a.items():
                    out[f"{tag}_{step}_{k}"] = v.numpy()
        print("packed" if packed else "dense", f"C={C}", "Gaussians per step:", sizes, "- identical to the reference after every step")
    path = os.path.join(ROOT, "tests", "golden", "strategy_ref.npz")
    np.savez_compressed(path, **out)
    print("STRATEGY PINNED ->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
