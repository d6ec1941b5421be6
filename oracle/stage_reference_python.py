"""Stage the reference's PURE-PYTHON package as an importable archive for the GPU-box drop-in test (TEST INFRASTRUCTURE).

`tests/test_gpu_reference_shim.py` drives the reference's own Python (gsplat.rasterization(), rasterization_2dgs(),
gsplat.strategy.DefaultStrategy, the autograd registration in gsplat/cuda/_wrapper.py) over this backend. The GPU box has no
reference checkout, so `__graft_entry__.build()` - which runs where /root/reference exists - calls this recipe: it zips the
`*.py` files of /root/reference/gsplat (no csrc, no binaries, nothing compiled) into oracle/_ref/reference_py.zip, next to the
oracle's other reference-derived build artefacts. oracle/_ref/ is git-ignored (reference sources never enter this repository's
history) and is not gpurun-ignored (it travels with the snapshot like the built .so files). Python imports packages straight
from a zip on sys.path. Nothing in the product path reads the archive; only the test does.
"""
import os
import sys
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref", "reference_py.zip")


def stage(reference_root: str = "/root/reference", out: str = OUT) -> str:
    pkg = os.path.join(reference_root, "gsplat")
    if not os.path.isdir(pkg):
        return ""
    os.makedirs(os.path.dirname(out), exist_ok=True)
    tmp = out + ".tmp"
    with zipfile.ZipFile(tmp, "w", zipfile.ZIP_DEFLATED) as z:
        for base, dirs, files in os.walk(pkg):
            dirs[:] = sorted(d for d in dirs if d not in ("__pycache__", "csrc", "third_party"))
            for f in sorted(files):
                if f.endswith(".py"):
                    full = os.path.join(base, f)
                    z.write(full, os.path.relpath(full, reference_root))
    os.replace(tmp, out)
    return out


TESTS_OUT = os.path.join(HERE, "_ref", "reference_tests.zip")


def stage_tests(reference_root: str = "/root/reference", out: str = TESTS_OUT) -> str:
    """The reference's own pytest tree for tools/run_reference_suite.py: the `*.py` and small data files under tests/ (no C++
    sources), the root conftest.py and assets/test_garden.npz. Same rules as `stage`: a git-ignored build artefact, read by
    the test runner only."""
    tests = os.path.join(reference_root, "tests")
    if not os.path.isdir(tests):
        return ""
    os.makedirs(os.path.dirname(out), exist_ok=True)
    tmp = out + ".tmp"
    with zipfile.ZipFile(tmp, "w", zipfile.ZIP_DEFLATED) as z:
        for base, dirs, files in os.walk(tests):
            dirs[:] = sorted(d for d in dirs if d not in ("__pycache__", "cpp"))
            for f in sorted(files):
                if f.endswith((".py", ".json", ".npz")):
                    full = os.path.join(base, f)
                    z.write(full, os.path.relpath(full, reference_root))
        for extra in ("conftest.py", os.path.join("assets", "test_garden.npz")):
            full = os.path.join(reference_root, extra)
            if os.path.exists(full):
                z.write(full, extra)
    os.replace(tmp, out)
    return out


if __name__ == "__main__":
    print(stage(*(sys.argv[1:2])) or "no reference checkout: nothing staged")
    print(stage_tests(*(sys.argv[1:2])) or "no reference checkout: no tests staged")
