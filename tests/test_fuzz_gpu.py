"""A short randomised GPU-vs-oracle sweep inside the test suite (the long runs are `python tools/fuzz_parity.py N SEED`):
random geometry, window position / width, reflectance, NODATA value, RGB bands, invalid pixels, starved columns, constant
bands -- status, valid-row counts, alpha indices, bgmeta, NODATA placement and RGB exact, scores and statistics 1e-4."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("seed,wide", [(1001, False), (1002, False), (1003, True)])
def test_random_parity_cases(seed, wide):
    import torch
    assert torch.cuda.is_available()
    import fuzz_parity
    assert fuzz_parity.run(ncase=10, seed=seed, WIDE=wide, verbose=False) == 0
