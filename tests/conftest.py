import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLDEN, "ref_scipy_paths.npz"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import np_oracle
    np_oracle.build()
    return np_oracle


@pytest.fixture(scope="session")
def gpu():
    """Initialises the HIP engine; fails loudly (no skip, no fallback) when the device or the
    in-tree libofl_hip.so is missing."""
    import oflibnumpy_amd as of
    of.native.ensure_device()
    return of
