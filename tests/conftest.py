import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# `oracle` must stay the PACKAGE (oracle/__init__.py) for every `from oracle import ...` of the tests: some tests execute a
# pin script (oracle/pin_*.py), which puts the oracle directory itself at the front of sys.path - a first import of `oracle` after
# that would find oracle/oracle.py instead. Importing the (empty) package here pins sys.modules["oracle"] whatever the test order.
import oracle  # noqa: E402, F401


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a ROCm GPU (MI355X); run with -m gpu")


@pytest.fixture(autouse=True)
def _seed():
    torch.manual_seed(42)
    yield


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    path = os.path.join(ROOT, "tests", "golden", "garden_quarter.npz")
    return dict(np.load(path))
