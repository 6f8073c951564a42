import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pixpro-with-opticalflow_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, PKG, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Golden vectors captured from the reference (tests/golden/make_golden.py) as torch tensors."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        a = z[k]
        if a.dtype.kind in "US":
            out[k] = str(a)
        elif a.ndim == 0:
            out[k] = a.item()
        else:
            out[k] = torch.from_numpy(a.copy())
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    return torch.cuda.is_available()


@pytest.fixture(scope="session")
def dev():
    if not has_gpu():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
