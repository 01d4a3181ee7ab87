import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def ref_goldens():
    return load_golden("reference_goldens.npz")


@pytest.fixture(scope="session")
def rbc_golden():
    return load_golden("rbc_linearized.npz")


@pytest.fixture(scope="session")
def sw_golden():
    return load_golden("sw_shaped.npz")


@pytest.fixture(scope="session")
def failure_golden():
    return load_golden("failure_cases.npz")


def has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False
