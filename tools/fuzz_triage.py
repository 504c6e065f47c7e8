#!/usr/bin/env python3
"""Random products through the column-profile kernels (plain and robust) against oracle/triage_oracle.py (numpy):
random line counts (not powers of two), columns without / with one / two valid positive pixels, ties, NODATA, NaN."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from srcfinder_amd import triage
from oracle import triage_oracle as T

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for case in range(40):
    L = int(rng.choice([1, 2, 3, 17, 100, 255, 256, 257, 1000, 4097, 20000, 32768, 32769, 50001]))   # > 32768: the radix-select kernel
    S = int(rng.choice([1, 5, 64, 65, 130]))
    img = rng.normal(200.0, 300.0, size=(L, S, 4))
    img[rng.random((L, S)) < 0.2, 3] = -9999.0
    if rng.random() < 0.5:
        img[..., 3] = np.round(img[..., 3] / 50.0) * 50.0          # many ties
        img[img[..., 3] == -10000.0, 3] = -9999.0
    for s in range(0, S, 7):
        img[:, s, 3] = -9999.0                                      # no valid pixel
    if S > 3:
        img[:, 3, 3] = -5.0; img[L // 2, 3, 3] = 7.0                # exactly one positive pixel
    a = triage.column_profile(img)
    b = T.column_profile(img[..., 3])
    # numpy reduces the float32 plane along the lines in float32, sequentially: at 20000 lines its own mean / std carry
    # ~2e-5 of rounding; the kernel accumulates the same float32 values in float64
    ok = np.allclose(a, b, rtol=1e-4, atol=1e-6, equal_nan=True) and np.array_equal(a[0], b[0], equal_nan=True)
    ok = ok and np.array_equal(a[3:], b[3:], equal_nan=True)          # min / max are exact
    ar = triage.column_profile(img, robust=True)
    br = T.column_profile_robust(img[..., 3])
    okr = np.array_equal(ar, br, equal_nan=True)
    if not (ok and okr):
        bad += 1
        print("MISMATCH case %d L %d S %d plain %s robust %s" % (case, L, S, ok, okr))
        if not ok:
            d = np.argwhere(~(np.isclose(a, b, rtol=1e-4, atol=1e-6) | (np.isnan(a) & np.isnan(b))))
            print("  plain diffs", [(tuple(x), a[tuple(x)], b[tuple(x)]) for x in d[:4]])
        if not okr:
            d = np.argwhere(~((ar == br) | (np.isnan(ar) & np.isnan(br))))
            print("  first diffs", d[:5].tolist(), ar[tuple(d[0])], br[tuple(d[0])])
print("fuzz triage: %d mismatches" % bad)
sys.exit(1 if bad else 0)
