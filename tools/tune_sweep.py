#!/usr/bin/env python3
"""Time stage 5 (the LOO sweep) alone on a full-size flightline for several kernel variants, interleaved rounds in one
process.  usage: tune_sweep.py [variants] [key]   (key 4 = sweep_variant, key 20 = sweep4_form; default key 20, forms 1,0,2)
Prints per variant the median / min time, whether the alpha indices equal the first variant's, and the largest relative
difference of the NLL curves (finite entries) against the first variant."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from srcfinder_amd import _ffi, cmf
from srcfinder_amd.synth import make_cube_torch

variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 0, 2]
key = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lines, samples, p, a0 = 20000, 598, 72, 351
lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
cube = make_cube_torch(lines, samples, seed=1, abscf_full=lib[:, 2])
L = _ffi.lib(); dev = cube.device; P = _ffi.ptr
al_np = cmf.alpha_grid(); na = len(al_np)
ws = torch.empty(L.sf_cmf_workspace_bytes(lines, p, samples, na), dtype=torch.uint8, device=dev)
f64 = dict(dtype=torch.float64, device=dev); i32 = dict(dtype=torch.int32, device=dev)
xt = torch.empty((samples, lines, p), dtype=torch.float32, device=dev)
mask = torch.empty((samples, lines), dtype=torch.uint8, device=dev)
nuse = torch.empty(samples, **i32); mu = torch.empty((samples, p), **f64); S = torch.empty((samples, p, p), **f64)
d = torch.empty((samples, p), **f64); lam = torch.empty((samples, p), **f64); evec = torch.empty((samples, p, p), **f64)
status = torch.empty(samples, **i32); nll = torch.empty((samples, na), **f64); aidx = torch.empty(samples, **i32)
al = torch.as_tensor(al_np, device=dev); st = _ffi.stream_ptr()
_ffi.check(L.sf_cmf_extract_columns(P(cube), lines, 425, samples, 0, samples, a0 - 1, p, P(xt), P(mask), st), "extract")
_ffi.check(L.sf_cmf_column_mean(P(xt), 0, P(mask), lines, p, samples, P(nuse), P(mu), P(ws), st), "mean")
_ffi.check(L.sf_cmf_covariance(P(xt), 0, P(mask), P(nuse), P(mu), lines, p, samples, P(S), P(ws), st), "cov")
_ffi.check(L.sf_cmf_eigh(P(S), P(nuse), p, samples, P(d), P(lam), P(evec), P(status), P(ws), st), "eigh")
def run():
    _ffi.check(L.sf_cmf_loocv(P(xt), 0, P(mask), P(nuse), P(mu), P(d), P(lam), P(evec), P(status), P(al), na,
                              lines, p, samples, P(nll), P(aidx), P(ws), st), "loocv")
ref = None; refnll = None
times = {v: [] for v in variants}
res = {}
for rnd in range(5):
    for v in variants:
        L.sf_debug_set(key, v)
        if rnd == 0:
            run(); torch.cuda.synchronize()
            res[v] = (aidx.cpu().numpy().copy(), nll.cpu().numpy().copy())
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize(); times[v].append(a.elapsed_time(b))
L.sf_debug_set(key, 0)
ref, refnll = res[variants[0]]
for v in variants:
    ai, nl = res[v]
    fin = np.isfinite(refnll) & np.isfinite(nl)
    same_pattern = np.array_equal(np.isfinite(refnll), np.isfinite(nl))
    rel = np.abs(nl[fin] - refnll[fin]) / np.maximum(np.abs(refnll[fin]), 1e-300)
    print("key %d variant %2d: median %.3f ms  min %.3f  alpha idx equal: %s  finite pattern equal: %s  max rel NLL diff %.3e  bit-identical NLL: %s"
          % (key, v, np.median(times[v]), min(times[v]), np.array_equal(ai, ref), same_pattern, rel.max() if rel.size else 0.0,
             np.array_equal(nl, refnll, equal_nan=True)))
