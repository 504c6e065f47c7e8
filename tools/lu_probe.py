#!/usr/bin/env python3
"""Blocked vs unblocked determinant kernel vs scipy on a ladder of sizes (debug aid for linalg.hip)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scipy.linalg as sla
from srcfinder_amd import cmf, _ffi

rng = np.random.default_rng(77)
for n in (1, 2, 3, 15, 16, 17, 18, 31, 32, 33, 48, 49, 100, 257, 425, 512, 600):
    a = rng.normal(size=(n, n))
    ref = sla.det(a)
    got = cmf.det(a)
    _ffi.lib().sf_debug_set(14, 1)
    old = cmf.det(a)
    _ffi.lib().sf_debug_set(14, 0)
    print(n, ref, got / ref if ref else got, old / ref if ref else old)
