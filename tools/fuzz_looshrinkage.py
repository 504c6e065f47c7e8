#!/usr/bin/env python3
"""Random calls of the function-level entries looshrinkage() / cov() (rows above / near / below the band count, n
different from the row count, optional full target I_reg, scaled data) against the faithful oracle.
python tools/fuzz_looshrinkage.py [cases=60] [seed=0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np
from srcfinder_amd import cmf
from srcfinder_amd.synth import synth_columns
from oracle import cmf_oracle as O

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
al = cmf.alpha_grid()
bad = 0
for case in range(ncase):
    p = int(rng.choice([2, 3, 8, 17, 40, 64, 69, 72, 72, 83, 96, 97, 130]))
    rows = int(max(2, rng.choice([p - 3, p, p + 1, p + 2, 2 * p, 5 * p, 20 * p])))
    n = rows if rng.random() < 0.6 else rows + int(rng.integers(1, 400))
    scale = float(rng.choice([1.0, 1.0, 1e-3, 50.0]))
    x = synth_columns(rows, p, int(rng.integers(1 << 30)), scale)
    x = x - x.mean(0)
    reg = ()
    if p <= 96 and rng.random() < 0.3:
        extra = synth_columns(rows + 50, p, int(rng.integers(1 << 30)), scale)
        reg = extra - x.mean(0)
    nll_o, nll_g = np.zeros(201), np.zeros(201)
    with np.errstate(all="ignore"):
        c_o, i_o = O.looshrinkage(x, al, nll_o, n, reg)
    c_g, i_g = cmf.looshrinkage(x, al, nll_g, n, reg)
    fo, fg = np.isfinite(nll_o), np.isfinite(nll_g)
    both = fo & fg
    ok = True
    why = ""
    # index: equal, or a numerical tie of the oracle's own curve
    if i_g != i_o:
        tie = i_g >= 0 and i_o >= 0 and np.isfinite(nll_o[i_g]) and abs(nll_o[i_g] - nll_o[i_o]) <= 1e-9 * abs(nll_o[i_o])
        ok, why = bool(tie), "index %d vs %d" % (i_g, i_o)
    # finite pattern may differ only at the over/underflow edges of det (a few grid points, contiguous, finite on the GPU side)
    dif = np.nonzero(fo != fg)[0]
    if len(dif) > 16 or (len(dif) and not (np.all(fg[dif]) or rows <= p + 1)):
        ok, why = False, why + " finite pattern %s" % dif[:8]
    if rows > p + 1 and both.any():
        # rows == p + 2: the matrices the reference inverts have a condition of 1e10 and more -- still checked, at the
        # looser bar that conditioning allows (the case used to be dropped altogether)
        r = np.abs(nll_g[both] - nll_o[both]) / np.maximum(np.abs(nll_o[both]), 1e-300)
        bar = 1e-7 if rows > p + 2 else 1e-3
        if r.max() > bar:
            ok, why = False, why + " nll rel %.2e (bar %.0e)" % (r.max(), bar)
    if i_g == i_o and rows > p + 1 and not np.allclose(c_g, c_o, rtol=1e-8, atol=1e-12 * np.abs(c_o).max()):
        ok, why = False, why + " C"
    cg, co = cmf.cov(x + 3.0), np.cov((x + 3.0).T)
    if not np.allclose(cg, np.atleast_2d(co), rtol=1e-9, atol=1e-12 * np.abs(co).max()):
        ok, why = False, why + " cov"
    if not ok:
        bad += 1
        print("MISMATCH case %d: p %d rows %d n %d scale %g reg %s: %s" % (case, p, rows, n, scale, len(reg) != 0, why))
print("fuzz looshrinkage: %d cases, %d mismatches" % (ncase, bad))
sys.exit(1 if bad else 0)
