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


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]

    return load


@pytest.fixture(scope="session")
def native():
    """The ctypes binding; building the library first if it is missing."""
    try:
        import torch  # noqa: F401  (tests of the resident forms use torch: it must be in the process BEFORE the library is loaded)
    except ImportError:
        pass
    from cvx_proj_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    _native.lib()
    return _native


def ulp_diff_f32(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)
