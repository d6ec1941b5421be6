"""CPU: the host side of rasterization() / rasterization_2dgs() on EMPTY scenes — orchestration, op bodies, ctypes
marshalling of every C-ABI call and the entry points' own argument checks — exercised without a GPU by tools/dry_run.py
(device launches are ignored there; the control flow of an empty scene does not depend on kernel results). The same
scenes run for real in tests/test_gpu_pipeline.py::test_rasterization_degenerate_scenes."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_empty_scenes_host_path():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dry_run.py")], capture_output=True, text=True,
                       cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count(": OK") == 4, r.stdout
