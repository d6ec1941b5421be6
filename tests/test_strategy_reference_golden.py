"""gsplat_amd.strategy.DefaultStrategy against the trajectory of the REFERENCE's gsplat/strategy (tests/golden/strategy_ref.npz,
written by oracle/pin_strategy_against_reference.py: the reference's own DefaultStrategy.step_post_backward driven on CPU through
statistics, three refinements and an opacity reset; dense and packed `info`). CPU: every parameter, both Adam moments of every
optimizer and the strategy state after the last step equal the reference's (the splits draw torch.randn in the same order).
GPU (marked): the same replay on the device - the model sizes after every step and everything that does not depend on the
device's RNG stream."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _replay(packed, C, device):
    from oracle import pin_strategy_against_reference as pin  # the generator's own scene / info / driver (the reference itself is
    #                                                           only imported inside its main())

    from gsplat_amd.strategy import DefaultStrategy

    params, opts = pin.make_model()
    if device != "cpu":
        params = torch.nn.ParameterDict({k: torch.nn.Parameter(v.detach().to(device)) for k, v in params.items()})
        moved = {}
        for k, o in opts.items():
            old = next(iter(o.state.values()))
            moved[k] = torch.optim.Adam([{"params": params[k], "lr": 1e-3, "name": k}])
            moved[k].state[params[k]] = {kk: (vv.to(device) if torch.is_tensor(vv) else vv) for kk, vv in old.items()}
        opts = moved
    strat = DefaultStrategy(**dict(pin.CFG, refine_scale2d_stop_iter=100 if C == 1 else 0))
    strat.check_sanity(params, opts)
    state = strat.initialize_state(scene_scale=1.0)
    torch.manual_seed(5)
    sizes = []
    for step in range(pin.STEPS):
        info = pin.make_info(len(params["means"]), step, packed, C)
        if device != "cpu":
            m2 = info["means2d"].detach().to(device).requires_grad_(True)
            m2.grad = info["means2d"].grad.to(device)
            info = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in info.items()}
            info["means2d"] = m2
        strat.step_post_backward(params, opts, state, step, info, packed=packed)
        sizes.append(len(params["means"]))
    return sizes, pin.snapshot(params, opts, state)


@pytest.mark.parametrize("packed,C", [(False, 2), (True, 2), (False, 1), (True, 1)])
def test_default_strategy_follows_the_reference_trajectory(packed, C):
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "strategy_ref.npz")))
    tag = f"{'packed' if packed else 'dense'}{C}"
    sizes, snap = _replay(packed, C, "cpu")
    assert sizes == g[f"{tag}_sizes"].tolist()
    last = len(sizes) - 1
    if f"{tag}_{last}_p_means" in g:  # the two trajectories whose final state the fixture keeps
        for k, v in snap.items():
            ref = torch.from_numpy(g[f"{tag}_{last}_{k}"])
            assert ref.shape == v.shape and torch.allclose(v, ref, rtol=1e-6, atol=1e-7), (tag, k)


@pytest.mark.gpu
@pytest.mark.parametrize("packed,C", [(False, 2), (True, 1)])
def test_default_strategy_follows_the_reference_trajectory_on_the_gpu(packed, C):
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "strategy_ref.npz")))
    tag = f"{'packed' if packed else 'dense'}{C}"
    sizes, snap = _replay(packed, C, "cuda")
    assert sizes == g[f"{tag}_sizes"].tolist()  # the same rows were cloned / split / pruned at every refinement
    last = len(sizes) - 1
    for k, v in snap.items():
        ref = torch.from_numpy(g[f"{tag}_{last}_{k}"])
        assert ref.shape == v.shape, (tag, k)
        if k in ("p_means", "m_means", "v_means"):
            continue  # positions of split children come from the device's RNG stream; their moments start at zero on both
        assert torch.allclose(v.cpu(), ref, rtol=1e-5, atol=1e-6), (tag, k)
    # rows that never descended from a split are untouched by the RNG: those positions are the reference's, bit for bit
    same = (snap["p_means"].cpu() == torch.from_numpy(g[f"{tag}_{last}_p_means"])).all(-1)
    assert int(same.sum()) > 0
