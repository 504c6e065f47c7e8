#!/usr/bin/env python3
"""Do a streaming kernel and the sweep share CUs when they are issued on two HIP streams?  Stage 5 (the LOO sweep, sf_cmf_loocv) on
stream A and stage 7 (the score kernel, sf_cmf_score) or stage 1 (extract) on stream B, each alone and both at once, for the sweep forms
given as key=value knob sets (default: the production kernel and round 2's four-wave form, one wave per SIMD).
usage: coresidency_probe.py ["20=1 8=4" ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from srcfinder_amd import _ffi, cmf
from srcfinder_amd.synth import make_cube_torch
lines, samples, p, a0 = 20000, 598, 72, 351
lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
cube = make_cube_torch(lines, samples, seed=1, abscf_full=lib[:, 2])
L = _ffi.lib(); dev = cube.device; P = _ffi.ptr
al_np = cmf.alpha_grid(); na = len(al_np)
f64 = dict(dtype=torch.float64, device=dev); i32 = dict(dtype=torch.int32, device=dev)
ws = torch.empty(L.sf_cmf_workspace_bytes(lines, p, samples, na), dtype=torch.uint8, device=dev)
ws2 = torch.empty(L.sf_cmf_workspace_bytes(lines, p, samples, na), dtype=torch.uint8, device=dev)
xt = torch.empty((samples, lines, p), dtype=torch.float32, device=dev); xt2 = torch.empty_like(xt)
mask = torch.empty((samples, lines), dtype=torch.uint8, device=dev); mask2 = torch.empty_like(mask)
nuse = torch.empty(samples, **i32); mu = torch.empty((samples, p), **f64); S = torch.empty((samples, p, p), **f64)
d = torch.empty((samples, p), **f64); lam = torch.empty((samples, p), **f64); evec = torch.empty((samples, p, p), **f64)
status = torch.empty(samples, **i32); nll = torch.empty((samples, na), **f64); aidx = torch.empty(samples, **i32)
al = torch.as_tensor(al_np, device=dev)
st0 = _ffi.stream_ptr()
_ffi.check(L.sf_cmf_extract_columns(P(cube), lines, 425, samples, 0, samples, a0 - 1, p, P(xt), P(mask), st0), "extract")
_ffi.check(L.sf_cmf_column_mean(P(xt), 0, P(mask), lines, p, samples, P(nuse), P(mu), P(ws), st0), "mean")
_ffi.check(L.sf_cmf_covariance(P(xt), 0, P(mask), P(nuse), P(mu), lines, p, samples, P(S), P(ws), st0), "cov")
_ffi.check(L.sf_cmf_eigh(P(S), P(nuse), p, samples, P(d), P(lam), P(evec), P(status), P(ws), st0), "eigh")
g = torch.Generator(device=dev); g.manual_seed(3)
filt = torch.randn((samples, p), generator=g, **f64); bias = torch.randn(samples, generator=g, **f64)
status0 = torch.zeros(samples, **i32); aidx0 = torch.full((samples,), 130, **i32); nuse0 = torch.full((samples,), lines, **i32)
out = torch.empty((lines, samples, 4), **f64); colstats = torch.empty((3, samples), **f64)
torch.cuda.synchronize()
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def sweep():
    with torch.cuda.stream(sA):
        _ffi.check(L.sf_cmf_loocv(P(xt), 0, P(mask), P(nuse), P(mu), P(d), P(lam), P(evec), P(status), P(al), na,
                                  lines, p, samples, P(nll), P(aidx), P(ws), sA.cuda_stream), "loocv")
def score():
    with torch.cuda.stream(sB):
        _ffi.check(L.sf_cmf_score(P(cube), lines, 425, samples, 0, samples, a0 - 1, p, P(filt), P(bias), P(status0), P(aidx0), P(nuse0),
                                  60, 42, 24, -9999.0, P(out), samples, 0, 4, None, P(colstats), P(ws2), sB.cuda_stream), "score")
def extract():
    with torch.cuda.stream(sB):
        _ffi.check(L.sf_cmf_extract_columns(P(cube), lines, 425, samples, 0, samples, a0 - 1, p, P(xt2), P(mask2), sB.cuda_stream), "extract")
def copy():     # a plain device copy (no LDS, few registers): 3.4 GB read + 3.4 GB written
    with torch.cuda.stream(sB):
        xt2.copy_(xt)
def timeit(fs, n=10):
    for f in fs: f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for f in fs: f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
knobsets = sys.argv[1:] or ["", "20=1 8=4"]
for ks in knobsets:
    for k in (4, 8, 20, 21): L.sf_debug_set(k, {8: 8}.get(k, 0))
    for kv in ks.split():
        k, v = kv.split("="); L.sf_debug_set(int(k), int(v))
    a, b, c = timeit([sweep]), timeit([score]), timeit([extract])
    ab, ac, abc = timeit([sweep, score]), timeit([sweep, extract]), timeit([sweep, score, extract])
    cp, acp = timeit([copy]), timeit([sweep, copy])
    print("knobs [%s]: stage 5 %.3f ms, score %.3f, extract %.3f alone; stage 5 + score on two streams %.3f (sum %.3f), stage 5 + extract %.3f (sum %.3f), "
          "all three %.3f (sum %.3f); a 3.4 GB device copy %.3f alone, stage 5 + copy %.3f (sum %.3f)" % (ks, a, b, c, ab, a + b, ac, a + c, abc, a + b + c, cp, acp, a + cp))
