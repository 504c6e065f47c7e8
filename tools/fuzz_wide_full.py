#!/usr/bin/env python3
"""Randomised parity sweep of the full-column shrinkage target on the WIDE path (-R -k 2|3 -f [-r], window 5..420, p = 416):
sf_cmf_wide_stats_target (blocked Cholesky of the target, substitution whitening, unit-mode block Jacobi, exact determinants
of n beta S + alpha T) against the faithful oracle, labels injected.   python tools/fuzz_wide_full.py [cases=8] [seed=0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "8")
import numpy as np
from srcfinder_amd import cmf
from srcfinder_amd.synth import make_cube_numpy
from oracle import cmf_oracle as O

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
bad = 0
t0 = time.time()
for case in range(ncase):
    lines = int(rng.choice([520, 700, 900]))
    samples = int(rng.choice([1, 2]))
    k = int(rng.choice([2, 3]))
    reject = bool(rng.random() < 0.4)
    cube = make_cube_numpy(lines, samples, seed=int(rng.integers(1 << 30)), abscf_full=lib[:, 2], active=(5, 420),
                           nodata_lines=int(rng.integers(0, 3)), nodata_column=-1)
    lab = np.zeros((lines, samples), np.int64)
    for s in range(samples):
        cuts = np.sort(rng.integers(40, lines - 40, size=k - 1))
        if rng.random() < 0.4:
            cuts[-1] = lines - int(rng.integers(20, 150))         # a small last cluster (fewer rows than bands: S singular, T not)
        lab[:, s] = rng.permutation(k)[np.searchsorted(cuts, np.arange(lines), side="right")]
    b0, b1 = int(lines * 0.3), int(lines * 0.6)
    cube[b0:b1] *= np.float32(1.0 + 0.4 * rng.random())
    g = cmf.robust_mf(cube, lib, reflectance=True, kmeans=k, labels=lab, reject=reject, full=True, metadata=True, to_numpy=True)
    with np.errstate(all="ignore"):
        o = O.robust_mf_multimodal_oracle(cube, lib, lab, reflectance=True, reject=reject, full=True)
    ok = np.array_equal(g.out[..., 3] == -9999.0, o["out"][..., 3] == -9999.0)
    ok = ok and np.array_equal(g.bgmeta, o["bgmeta"])
    a, b = g.out[..., 3], o["out"][..., 3]
    fin = np.isfinite(b) & (b != -9999.0)
    ok = ok and np.array_equal(np.isnan(a), np.isnan(b))
    worst = float((np.abs(a[fin] - b[fin]) / (1e-4 * np.abs(b[fin]) + 1e-7 * np.abs(b[fin]).max())).max()) if fin.any() else 0.0
    ok = ok and worst <= 1.0
    sizes = [[int((lab[:, s] == q).sum()) for q in range(k)] for s in range(samples)]
    print("case %d: lines %d samples %d k %d reject %s sizes %s alpha idx %s worst %.3g %s" %
          (case, lines, samples, k, reject, sizes, np.unique(o["bgmeta"][..., 1]).tolist(), worst, "ok" if ok else "MISMATCH"), flush=True)
    bad += 0 if ok else 1
print("fuzz wide full target: %d cases, %d mismatches (%.0f s)" % (ncase, bad, time.time() - t0))
