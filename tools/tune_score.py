#!/usr/bin/env python3
"""A/B the score kernel's launch shape on a full-size cube (interleaved rounds in ONE process)."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from srcfinder_amd import _ffi, cmf
from srcfinder_amd.synth import make_cube_torch

lines, samples, p = 20000, 598, 72
lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
cube = make_cube_torch(lines, samples, seed=1, abscf_full=lib[:, 2])
L = _ffi.lib()
dev = cube.device
g = torch.Generator(device=dev); g.manual_seed(3)
filt = torch.randn((samples, p), dtype=torch.float64, device=dev, generator=g)
bias = torch.randn(samples, dtype=torch.float64, device=dev, generator=g)
status = torch.zeros(samples, dtype=torch.int32, device=dev)
aidx = torch.full((samples,), 130, dtype=torch.int32, device=dev)
nuse = torch.full((samples,), lines, dtype=torch.int32, device=dev)
out = torch.empty((lines, samples, 4), dtype=torch.float64, device=dev)
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
P = _ffi.ptr

def run():
    _ffi.check(L.sf_cmf_score(P(cube), lines, 425, samples, 0, samples, 350, p, P(filt), P(bias), P(status), P(aidx), P(nuse),
                              60, 42, 24, -9999.0, P(out), samples, 0, 4, None, None, P(ws), _ffi.stream_ptr()), "score")

cfgs = []
for variant, lpw, xcd in itertools.product([0], [16, 24, 32, 40, 48, 64, 80], [1]):
    cfgs.append((variant, lpw, xcd))
res = {c: [] for c in cfgs}
for rnd in range(4):
    for c in cfgs:
        L.sf_debug_set(1, c[0]); L.sf_debug_set(2, c[1]); L.sf_debug_set(3, c[2])
        run(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize()
        res[c].append(a.elapsed_time(b))
names = {0: "8x4", 1: "2x16", 3: "4x4", 4: "2x8", 5: "8x8", 6: "4x8", 7: "8x4/128", 8: "4x8/128", 9: "4x4/128"}
L.sf_debug_set(1, 0); L.sf_debug_set(2, 0); run(); torch.cuda.synchronize(); ref = out.clone()
for v in (7, 8, 9):
    L.sf_debug_set(1, v); out.zero_(); run(); torch.cuda.synchronize(); print("variant", v, "bit-identical to default:", bool(torch.equal(out, ref)))
for c in sorted(cfgs, key=lambda c: np.median(res[c])):
    print("LPIxUB %-5s lpw %4d xcd %d : median %.3f ms min %.3f  -> %.0f GB/s" % (names[c[0]], c[1], c[2], np.median(res[c]), min(res[c]), 332 * lines * samples / np.median(res[c]) / 1e6))
