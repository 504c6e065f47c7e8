#!/usr/bin/env python3
"""Time stage 3 (covariance) alone on a full-size flightline for sf_debug_set(5, v) variants and compare results."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from srcfinder_amd import _ffi, cmf
from srcfinder_amd.synth import make_cube_torch

variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 2, 1]
lines, samples, p, a0 = 20000, 598, 72, 351
lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
cube = make_cube_torch(lines, samples, seed=1, abscf_full=lib[:, 2])
L = _ffi.lib(); dev = cube.device; P = _ffi.ptr
ws = torch.empty(L.sf_cmf_workspace_bytes(lines, p, samples, 201), dtype=torch.uint8, device=dev)
f64 = dict(dtype=torch.float64, device=dev); i32 = dict(dtype=torch.int32, device=dev)
xt = torch.empty((samples, lines, p), dtype=torch.float32, device=dev)
mask = torch.empty((samples, lines), dtype=torch.uint8, device=dev)
nuse = torch.empty(samples, **i32); mu = torch.empty((samples, p), **f64); S = torch.empty((samples, p, p), **f64)
st = _ffi.stream_ptr()
_ffi.check(L.sf_cmf_extract_columns(P(cube), lines, 425, samples, 0, samples, a0 - 1, p, P(xt), P(mask), st), "extract")
_ffi.check(L.sf_cmf_column_mean(P(xt), 0, P(mask), lines, p, samples, P(nuse), P(mu), P(ws), st), "mean")
def run():
    _ffi.check(L.sf_cmf_covariance(P(xt), 0, P(mask), P(nuse), P(mu), lines, p, samples, P(S), P(ws), st), "cov")
ref = None
for v in variants:
    L.sf_debug_set(5, v)
    S.zero_(); run(); torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    Sn = S.cpu().numpy()
    if ref is None: ref = Sn
    print("variant %2d: median %.3f ms  min %.3f  max rel diff to first %.3e  symmetric %s" % (
        v, np.median(ts), min(ts), np.abs(Sn - ref).max() / np.abs(ref).max(), np.array_equal(Sn, Sn.transpose(0, 2, 1))))
