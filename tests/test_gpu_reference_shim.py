"""GPU + reference checkout: the REFERENCE's own Python (gsplat.rasterization(), gsplat.strategy.DefaultStrategy, its
autograd registration in gsplat/cuda/_wrapper.py) driven over this backend through the `gsplat.csrc` shim
(INTEGRATION.md route A) - the drop-in claim exercised end to end on hardware.

The reference sources are not part of this repository (and must not be copied into it): the test looks for a checkout at
$GSPLAT_REFERENCE_PATH (default /root/reference) and is skipped when there is none. On a GPU box: mount or clone
nerfstudio-project/gsplat 1.6.0 there (sources only, nothing is compiled - the CUDA build is never attempted because
`from gsplat import csrc` succeeds first, gsplat/cuda/_backend.py:29-31) and run `pytest tests/test_gpu_reference_shim.py`."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("GSPLAT_REFERENCE_PATH", "/root/reference")

_SCRIPT = r'''
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r); sys.path.insert(0, %(ref)r)
import torch
import gsplat_amd.csrc_shim as shim
sys.modules["gsplat.csrc"] = shim            # what a one-line gsplat/csrc.py does (INTEGRATION.md)
import gsplat                                 # the reference package, unmodified
from gsplat.cuda._backend import _C
assert _C is shim
from _util import make_scene
import gsplat_amd

dev = "cuda"
sc, W, H = make_scene(N=4000, C=2, width=160, height=112, seed=4, sh_degree=3)
a = {k: v.to(dev) for k, v in sc.items()}
names = ("means", "quats", "scales", "opacities", "colors")

def run(fn):
    leaves = {k: a[k].clone().requires_grad_(True) for k in names}
    rc, ra, meta = fn(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"],
                      a["viewmats"], a["Ks"], W, H, sh_degree=3, packed=False)
    (rc.sum() + ra.sum()).backward()
    return rc.detach(), ra.detach(), {k: leaves[k].grad for k in names}

rc_ref, ra_ref, g_ref = run(gsplat.rasterization)       # reference orchestration + reference autograd over our ops
rc_own, ra_own, g_own = run(gsplat_amd.rasterization)   # this package's own orchestration
assert torch.allclose(rc_ref, rc_own, rtol=1e-4, atol=1e-5), float((rc_ref - rc_own).abs().max())
assert torch.allclose(ra_ref, ra_own, rtol=1e-4, atol=1e-5)
for k in names:
    s = float(g_own[k].abs().max()) + 1e-30
    assert float((g_ref[k] - g_own[k]).abs().max()) <= 2e-4 * s, k

# a short fit with the REFERENCE's DefaultStrategy editing the model between steps
from gsplat.strategy import DefaultStrategy
params = torch.nn.ParameterDict({
    "means": torch.nn.Parameter(a["means"].clone()), "quats": torch.nn.Parameter(a["quats"].clone()),
    "scales": torch.nn.Parameter(torch.log(a["scales"])), "opacities": torch.nn.Parameter(torch.logit(a["opacities"])),
    "sh": torch.nn.Parameter(a["colors"].clone())}).to(dev)
opts = {k: torch.optim.Adam([p], lr=1e-3) for k, p in params.items()}
strategy = DefaultStrategy(refine_start_iter=5, refine_every=5, reset_every=10_000, grow_grad2d=1e-6, verbose=False)
strategy.check_sanity(params, opts)
state = strategy.initialize_state(scene_scale=1.0)
target = rc_own.clamp(0, 1)
sizes, losses = set(), []
for step in range(16):
    rc, ra, info = gsplat.rasterization(params["means"], params["quats"], torch.exp(params["scales"]),
                                        torch.sigmoid(params["opacities"]), params["sh"], a["viewmats"], a["Ks"], W, H,
                                        sh_degree=3, packed=False)
    loss = (rc - target).abs().mean()
    strategy.step_pre_backward(params, opts, state, step, info)
    loss.backward()
    for o in opts.values():
        o.step(); o.zero_grad(set_to_none=True)
    strategy.step_post_backward(params, opts, state, step, info, packed=False)
    sizes.add(len(params["means"])); losses.append(float(loss))
assert len(sizes) > 1, "the reference strategy never edited the model"
assert all(l == l for l in losses)
print("OK", sorted(sizes), losses[0], losses[-1])
'''


def test_reference_python_runs_over_the_shim():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    if not os.path.isdir(os.path.join(REF, "gsplat")):
        pytest.skip(f"no reference checkout at {REF} (set GSPLAT_REFERENCE_PATH; see the module docstring)")
    code = _SCRIPT % {"root": ROOT, "tests": os.path.join(ROOT, "tests"), "ref": REF}
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp", env=env, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-3000:]
