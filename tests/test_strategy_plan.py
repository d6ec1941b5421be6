"""CPU: the one-pass refinement plans of gsplat_amd.strategy (RowPlan) against the reference's staged edits
(duplicate -> split -> remove, restated in strategy/ops.py with the reference's call signatures): same set of rows, same
inherited / fresh optimizer moments, statistics reset. Only the sampled positions of split children are random."""
import torch

from gsplat_amd.strategy import DefaultStrategy, ops


def _model(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    raw = dict(means=torch.randn(n, 3, generator=g), scales=torch.log(torch.rand(n, 3, generator=g) * 0.05 + 0.001),
               quats=torch.randn(n, 4, generator=g), opacities=torch.logit(torch.rand(n, generator=g) * 0.9 + 0.002),
               sh0=torch.randn(n, 1, 3, generator=g))
    params = torch.nn.ParameterDict({k: torch.nn.Parameter(v.clone()) for k, v in raw.items()})
    opts = {k: torch.optim.Adam([p], lr=1e-3) for k, p in params.items()}
    for k, p in params.items():
        p.grad = torch.randn(p.shape, generator=g)
        opts[k].step()
        p.grad = None
    return params, opts


def _rows(params, opts):
    """One sortable signature per Gaussian from the deterministic columns: opacity, scales, quats, sh0 + their moments."""
    cols = []
    for k in ("opacities", "scales", "quats", "sh0"):
        p = params[k]
        cols += [p.detach().reshape(len(p), -1), opts[k].state[p]["exp_avg"].reshape(len(p), -1),
                 opts[k].state[p]["exp_avg_sq"].reshape(len(p), -1)]
    m = torch.cat(cols, dim=1)
    order = sorted(range(len(m)), key=lambda i: m[i].tolist())
    return m[order]


def test_default_refinement_plan_equals_staged_edits():
    n = 300
    strategy = DefaultStrategy(refine_start_iter=0, refine_every=1, grow_grad2d=0.5, reset_every=2, prune_opa=0.05,
                               prune_scale3d=0.03, revised_opacity=True)
    g = torch.Generator().manual_seed(5)
    grad2d, count = torch.rand(n, generator=g), torch.ones(n)
    step = 5  # > reset_every: the size test of the pruning is active

    # staged, as the reference's DefaultStrategy does it
    p1, o1 = _model(n)
    st1 = {"grad2d": grad2d.clone(), "count": count.clone(), "scene_scale": 1.0}
    hot = st1["grad2d"] / st1["count"].clamp_min(1) > strategy.grow_grad2d
    small = torch.exp(p1["scales"]).max(-1).values <= strategy.grow_scale3d
    dup, spl = hot & small, hot & ~small
    ops.duplicate(p1, o1, st1, dup)
    spl = torch.cat([spl, torch.zeros(int(dup.sum()), dtype=torch.bool)])
    ops.split(p1, o1, st1, spl, revised_opacity=True)
    prune = torch.sigmoid(p1["opacities"].flatten()) < strategy.prune_opa
    prune |= torch.exp(p1["scales"]).max(-1).values > strategy.prune_scale3d
    ops.remove(p1, o1, st1, prune)

    # planned
    p2, o2 = _model(n)
    st2 = strategy.initialize_state(1.0)
    st2["grad2d"], st2["count"] = grad2d.clone(), count.clone()
    n_clone, n_split, n_prune = strategy._refine(p2, o2, st2, step)
    assert (n_clone, n_split, n_prune) == (int(dup.sum()), int(spl.sum()), int(prune.sum()))
    assert len(p2["means"]) == len(p1["means"]) and len(p2["means"]) != n
    torch.testing.assert_close(_rows(p2, o2), _rows(p1, o1), rtol=1e-6, atol=1e-7)
    for k, p in p2.items():  # bookkeeping: one parameter per optimizer, state keyed by the new tensor
        assert o2[k].param_groups[0]["params"][0] is p and o2[k].state[p]["exp_avg"].shape == p.shape
    assert st2["grad2d"].shape == (len(p2["means"]),) and float(st2["grad2d"].abs().sum()) == 0.0
    # children of a split sit within a few sigma of their parent (positions are the only random part)
    assert torch.isfinite(p2["means"]).all()


def test_row_plan_select_renumbers_overrides():
    plan = ops.RowPlan(torch.tensor([0, 1, 1, 2, 2]), torch.tensor([False, False, True, True, True]))
    plan.set("x", torch.tensor([2, 4]), torch.tensor([[10.0], [20.0]]))
    kept = plan.select(torch.tensor([True, False, True, False, True]))
    assert kept.src.tolist() == [0, 1, 2] and kept.fresh.tolist() == [False, True, True]
    rows, vals = kept.values["x"]
    assert rows.tolist() == [1, 2] and vals.flatten().tolist() == [10.0, 20.0]
    params = torch.nn.ParameterDict({"x": torch.nn.Parameter(torch.tensor([[1.0], [2.0], [3.0]]))})
    opts = {"x": torch.optim.Adam([params["x"]], lr=0.1)}
    params["x"].grad = torch.ones(3, 1)
    opts["x"].step()
    state = {"stat": torch.tensor([7.0, 8.0, 9.0]), "scalar": 3.0}
    ops.apply_plan(params, opts, state, kept)
    assert params["x"].flatten().tolist()[1:] == [10.0, 20.0] and abs(float(params["x"][0]) - 0.9) < 1e-6
    m = opts["x"].state[params["x"]]["exp_avg"].flatten().tolist()
    assert m[0] != 0.0 and m[1:] == [0.0, 0.0]
    assert state["stat"].tolist() == [7.0, 8.0, 9.0] and state["scalar"] == 3.0

