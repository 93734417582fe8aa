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


def pytest_sessionstart(session):
    """The built libraries are git-ignored: on a fresh checkout compile them once (hipcc cross-compiles
    gfx950 without a GPU; gcc builds the oracle).  On the GPU box the prebuilt in-tree files are used."""
    from oflibnumpy_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        from oflibnumpy_amd.build_native import build_native
        build_native()
    from oracle import np_oracle
    np_oracle.build()


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLDEN, "ref_scipy_paths.npz"))


@pytest.fixture(scope="session")
def golden2():
    """Round-2 reference outputs (tests/golden/make_golden.py::main_delaunay): folds, holes, curved borders, shear."""
    return np.load(os.path.join(GOLDEN, "ref_delaunay_cases.npz"))


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
