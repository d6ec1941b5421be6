"""GPU + the reference's Python: the REFERENCE's own Python (gsplat.rasterization(), rasterization_2dgs(),
gsplat.strategy.DefaultStrategy, its autograd registration in gsplat/cuda/_wrapper.py) driven over this backend through the
`gsplat.csrc` shim (INTEGRATION.md route A) - the drop-in claim exercised end to end on hardware, on the reference's DEFAULT
layout (packed=True) and on dense rows, RGB+ED, absgrad, 2DGS and distributed=True in a 1-rank RCCL group.

The reference sources are not part of this repository. The test looks for them, in this order, at $GSPLAT_REFERENCE_PATH, at
/root/reference (the build container), and in oracle/_ref/reference_py.zip - the archive of the reference's *.py files that
`__graft_entry__.build()` stages (oracle/stage_reference_python.py; git-ignored build artefact that travels to the GPU box
like the built libraries; Python imports packages straight from a zip). Nothing is compiled: `from gsplat import csrc`
succeeds first (gsplat/cuda/_backend.py:29-31), so the CUDA build is never attempted."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reference_path():
    for cand in (os.environ.get("GSPLAT_REFERENCE_PATH"), "/root/reference"):
        if cand and os.path.isdir(os.path.join(cand, "gsplat")):
            return cand
    staged = os.path.join(ROOT, "oracle", "_ref", "reference_py.zip")
    return staged if os.path.exists(staged) else None


REF = _reference_path()

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

# the reference's DEFAULT layout (packed=True, gsplat/rendering.py:252), RGB+ED, absgrad - same comparison
def compare(tag, ref_out, own_out, tol=2e-4):
    for i, (x, y) in enumerate(zip(ref_out[:-1], own_out[:-1])):
        assert torch.allclose(x, y, rtol=1e-4, atol=1e-5), (tag, i, float((x - y).abs().max()))
    for k in ref_out[-1]:
        gr, go = ref_out[-1][k], own_out[-1][k]
        gr = gr.to_dense() if gr.is_sparse else gr
        go = go.to_dense() if go.is_sparse else go
        s = float(go.abs().max()) + 1e-30
        assert float((gr - go).abs().max()) <= tol * s, (tag, k)

def run_kw(fn, **kw):
    leaves = {k: a[k].clone().requires_grad_(True) for k in names}
    rc, ra, meta = fn(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"],
                      a["viewmats"], a["Ks"], W, H, sh_degree=3, **kw)
    (rc.sum() + ra.sum()).backward()
    out = [rc.detach(), ra.detach()]
    if kw.get("absgrad"):
        out.append(meta["means2d"].absgrad.detach())
    return out + [{k: leaves[k].grad for k in names}]

for tag, kw in (("packed (the default)", {}), ("packed explicit", dict(packed=True)),
                ("packed RGB+ED", dict(packed=True, render_mode="RGB+ED")), ("dense RGB+ED", dict(packed=False, render_mode="RGB+ED")),
                ("dense absgrad", dict(packed=False, absgrad=True)), ("packed absgrad", dict(packed=True, absgrad=True)),
                ("packed sparse_grad", dict(packed=True, sparse_grad=True)),
                ("antialiased", dict(packed=False, rasterize_mode="antialiased")),
                ("segmented", dict(packed=False, segmented=True))):
    ref_out, own_out = run_kw(gsplat.rasterization, **kw), run_kw(gsplat_amd.rasterization, **kw)
    compare(tag, ref_out, own_out)
    if "sparse_grad" in kw:
        # (means also receives the dense view-direction gradient of the SH colours: sparse + dense accumulates dense)
        g_sp = run_kw(gsplat.rasterization, **kw)[-1]
        assert g_sp["quats"].is_sparse and g_sp["scales"].is_sparse
print("3DGS configurations OK")

# rasterization_2dgs through the reference's Python (RGB+ED, distortion loss, packed and dense)
def run_2dgs(fn, **kw):
    leaves = {k: a[k].clone().requires_grad_(True) for k in names}
    rc, ra, rn, rnd, rd, rm, meta = fn(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"],
                                       a["viewmats"], a["Ks"], W, H, sh_degree=3, **kw)
    (rc.sum() + ra.sum() + rn.sum() + rd.sum()).backward()
    return [rc.detach(), ra.detach(), rn.detach(), rd.detach()] + [{k: leaves[k].grad for k in names}]

for tag, kw in (("2dgs dense", dict(packed=False, render_mode="RGB+ED", distloss=True)),
                ("2dgs packed", dict(packed=True, render_mode="RGB+ED", distloss=True)),
                ("2dgs RGB", dict(packed=False))):
    compare(tag, run_2dgs(gsplat.rasterization_2dgs, **kw), run_2dgs(gsplat_amd.rasterization_2dgs, **kw), tol=5e-4)
print("2DGS configurations OK")

# distributed=True in a 1-rank RCCL group through the reference's Python (gsplat/rendering.py:178-198 hands the default NCCL
# group's name to the op): must equal the local render (reference tests/test_rasterization.py:819-868)
import os, torch.distributed as dist
import socket
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); free_port = sk.getsockname()[1]
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(free_port)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
try:
    for packed in (False, True):
        compare(f"distributed packed={packed}", run_kw(gsplat.rasterization, packed=packed, distributed=True),
                run_kw(gsplat_amd.rasterization, packed=packed, distributed=False))
finally:
    dist.destroy_process_group()
print("distributed 1-rank OK")

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
    if REF is None:
        pytest.skip("no reference Python found (GSPLAT_REFERENCE_PATH, /root/reference, oracle/_ref/reference_py.zip: "
                    "__graft_entry__.build() stages the archive where a reference checkout exists)")
    code = _SCRIPT % {"root": ROOT, "tests": os.path.join(ROOT, "tests"), "ref": REF}
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp", env=env, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-3000:]
