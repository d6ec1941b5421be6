import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


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
