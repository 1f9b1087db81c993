import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tiny-faces-pytorch_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def templates(golden):
    return golden("targets")["templates"]


def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def hip():
    """The ctypes-bound C-ABI library (tiny-faces-pytorch_amd/tinyfaces/_hip.py)."""
    from tinyfaces import _hip
    return _hip
