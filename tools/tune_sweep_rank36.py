#!/usr/bin/env python3
"""Time stage 5 on a flightline whose columns ALL need the rank-36 factorisation (correlation spectra over ~3.5 decades):
sf_debug_set(20, 3) = k_sweep4r<0,4,9> (one wave per SIMD, round 2), 0 = k_sweep4s<9> (two waves per SIMD, round 3, the default).
usage: tune_sweep_rank36.py [columns=128]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from srcfinder_amd import _ffi, cmf

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lines, p = 20000, 72
lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
g = torch.Generator(device="cuda"); g.manual_seed(5)
xt = torch.empty((ncol, lines, p), dtype=torch.float32, device="cuda")
for c in range(ncol):
    q, _ = torch.linalg.qr(torch.randn((p, p), generator=g, device="cuda", dtype=torch.float64))
    sd = torch.sqrt(torch.exp(torch.linspace(0.0, float(np.log(3e-4)), p, device="cuda", dtype=torch.float64)))
    xt[c] = (10.0 + 0.5 * (torch.randn((lines, p), generator=g, device="cuda", dtype=torch.float64) * sd) @ q.T).float()
L = _ffi.lib(); P = _ffi.ptr; dev = xt.device
al_np = cmf.alpha_grid(); na = len(al_np)
ws = torch.empty(L.sf_cmf_workspace_bytes(lines, p, ncol, na), dtype=torch.uint8, device=dev)
f64 = dict(dtype=torch.float64, device=dev); i32 = dict(dtype=torch.int32, device=dev)
mask = torch.ones((ncol, lines), dtype=torch.uint8, device=dev)
nuse = torch.empty(ncol, **i32); mu = torch.empty((ncol, p), **f64); S = torch.empty((ncol, p, p), **f64)
d = torch.empty((ncol, p), **f64); lam = torch.empty((ncol, p), **f64); evec = torch.empty((ncol, p, p), **f64)
status = torch.empty(ncol, **i32); nll = torch.empty((ncol, na), **f64); aidx = torch.empty(ncol, **i32)
al = torch.as_tensor(al_np, device=dev); st = _ffi.stream_ptr()
_ffi.check(L.sf_cmf_column_mean(P(xt), 0, P(mask), lines, p, ncol, P(nuse), P(mu), P(ws), st), "mean")
_ffi.check(L.sf_cmf_covariance(P(xt), 0, P(mask), P(nuse), P(mu), lines, p, ncol, P(S), P(ws), st), "cov")
_ffi.check(L.sf_cmf_eigh(P(S), P(nuse), p, ncol, P(d), P(lam), P(evec), P(status), P(ws), st), "eigh")
def run():
    _ffi.check(L.sf_cmf_loocv(P(xt), 0, P(mask), P(nuse), P(mu), P(d), P(lam), P(evec), P(status), P(al), na,
                              lines, p, ncol, P(nll), P(aidx), P(ws), st), "loocv")
res = {}
for rnd in range(4):
    for v in (3, 0):
        L.sf_debug_set(20, v)
        run(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize()
        res.setdefault(v, []).append(a.elapsed_time(b))
        if rnd == 0: res[(v, "out")] = (aidx.cpu().numpy().copy(), nll.cpu().numpy().copy())
L.sf_debug_set(20, 0)
a0, n0 = res[(3, "out")]; a3, n3 = res[(0, "out")]
print("%d columns x %d lines, all rank 36: k_sweep4r<0,4,9> %.3f ms, k_sweep4s<9> %.3f ms per stage-5 call; alpha idx equal %s (%s), NLL bit-identical %s"
      % (ncol, lines, np.median(res[3]), np.median(res[0]), np.array_equal(a0, a3), sorted(set(a0.tolist())), np.array_equal(n0, n3, equal_nan=True)))
