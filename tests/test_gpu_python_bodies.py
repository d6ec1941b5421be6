"""GPU: the PYTHON bodies of the ops that csrc/torch_ops.cpp also compiles (gsplat_amd/_ops.py; selected by
GSPLAT_AMD_COMPILED_OPS=0 and whenever an A/B build of the kernel library is loaded through GSPLAT_AMD_LIB). The default suite
runs the compiled bodies; this runs the stage, segment and pipeline suites once more over the Python ones in a subprocess, so
that the duplicate marshalling cannot rot unseen."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_PROBE = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import torch, gsplat_amd
from gsplat_amd import _ops
from _util import make_scene
names = []
real = _ops.call
_ops.call = lambda name, *a: (names.append(name), real(name, *a))[1]
sc, W, H = make_scene(N=2000, C=1, width=96, height=64, seed=1)
d = {k: v.to("cuda").requires_grad_(k in ("means", "colors")) for k, v in sc.items()}
rc, ra, meta = gsplat_amd.rasterization(d["means"], d["quats"], d["scales"], d["opacities"], d["colors"], d["viewmats"], d["Ks"], W, H,
                                        packed=False)
rc.sum().backward()
# the FORWARD entry points: the package's own autograd always runs the Python bodies of the backward ops (_autograd._bwd)
need = {"gsx_project_ewa_fwd", "gsx_raster3d_fwd"}
assert need <= set(names), sorted(need - set(names))
print("PYTHON BODIES", len(names))
"""


def test_python_op_bodies_run_the_suites():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    env = dict(os.environ, GSPLAT_AMD_COMPILED_OPS="0", PYTHONDONTWRITEBYTECODE="1")
    probe = subprocess.run([sys.executable, "-c", _PROBE % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}],
                           capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert probe.returncode == 0 and "PYTHON BODIES" in probe.stdout, probe.stderr[-3000:]
    # with the compiled bodies (the default) the same probe must NOT see those entry points go through ctypes
    env_c = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env_c.pop("GSPLAT_AMD_COMPILED_OPS", None)
    probe_c = subprocess.run([sys.executable, "-c", _PROBE % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}],
                             capture_output=True, text=True, cwd=ROOT, env=env_c, timeout=600)
    assert probe_c.returncode != 0 and "AssertionError" in probe_c.stderr, "the probe cannot tell the two bodies apart"
    sel = ["tests/test_gpu_ops.py", "tests/test_gpu_segments.py", "tests/test_gpu_pipeline.py", "tests/test_gpu_composite.py",
           "tests/test_gpu_sparse.py"]
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", *sel, "-k",
                          "not c2_garden and not c3_matches and not c4_matches and not full_size"],
                         capture_output=True, text=True, cwd=ROOT, env=env, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
