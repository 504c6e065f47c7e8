#!/usr/bin/env python3
"""Debug aid: a 40-row cluster with the full-column shrinkage target (p = 72): GPU looshrinkage against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from srcfinder_amd import cmf
from oracle import cmf_oracle as O
d = np.load(os.path.join(ROOT, "tests", "golden", "_tmp_case.npz"))
x, valid, lab = d["x"], d["valid"], d["lab"]
rows = valid & (lab == 0)
mu = x[rows].mean(0)
al = cmf.alpha_grid()
n = int(valid.sum())
no, ng = np.zeros(201), np.zeros(201)
with np.errstate(all="ignore"):
    Co, io = O.looshrinkage(x[rows] - mu, al, no, n, x[valid] - mu)
Cg, ig = cmf.looshrinkage(x[rows] - mu, al, ng, n, x[valid] - mu)
print("oracle idx", io, "gpu idx", ig)
np.set_printoptions(linewidth=200, precision=6)
print("oracle", no[8:22])
print("gpu   ", ng[8:22])
print("diff  ", (ng - no)[8:22])
print("diff tail", (ng - no)[100::20])
# the pieces: S eigen-structure through the generalised problem
import scipy.linalg as sla
S = np.cov((x[rows] - mu).T * 100.0)
T = np.cov((x[valid] - mu).T * 100.0)
w = sla.eigh(S, T, eigvals_only=True)
print("generalised eigenvalues: min %.3g max %.3g, #<1e-12: %d" % (w.min(), w.max(), (w < 1e-12 * w.max()).sum()))
