import os
import sys

os.environ.setdefault("OMP_NUM_THREADS", "1")   # tiny LAPACK problems: 1 thread is ~100x faster here

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import pytest

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def library():
    return np.load(os.path.join(GOLDEN, "ch4_library.npz"))["library"]


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def score_close(got, ref, rel=1e-4):
    """The parity metric of SURVEY.md §7.3: |d| <= rel*|ref| + 1e-9*max|ref| elementwise."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    tol = rel * np.abs(ref) + 1e-9 * np.max(np.abs(ref)) if ref.size else 0.0
    return np.abs(got - ref) <= tol
