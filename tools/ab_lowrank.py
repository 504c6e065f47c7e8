#!/usr/bin/env python3
"""Dump the rank factorisation (k_lowrank through sf_debug_lowrank) of a set of spectra to an .npz and time the launch:
   ab_lowrank.py out.npz   -- run once per library build, then compare the files (tools/ab_lowrank.py --cmp a.npz b.npz)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        same = np.array_equal(a[k], b[k], equal_nan=True) if a[k].dtype.kind == "f" else np.array_equal(a[k], b[k])
        print(k, "identical" if same else "DIFFERENT (max abs diff %.3e)" % np.nanmax(np.abs(a[k].astype(float) - b[k].astype(float))))
    sys.exit(0)
import torch
from srcfinder_amd import _ffi, cmf
L = _ffi.lib(); P = _ffi.ptr
p, nc = 72, 598
rng = np.random.default_rng(5)
lam = np.empty((nc, p))
for c in range(nc):
    dec = [2.5, 3.5, 4.5, 5.5, 6.5, 8.0][c % 6]          # eigenvalue range in decades: ranks 28, 36 and the full-rank verdict
    e = 10.0 ** (-dec * np.sort(rng.random(p)))
    lam[c] = e * p / e.sum()
al = cmf.alpha_grid(); na = len(al)
dev = "cuda"
lam_d = torch.as_tensor(lam, device=dev); al_d = torch.as_tensor(al, device=dev)
nuse = torch.full((nc,), 20000, dtype=torch.int32, device=dev); status = torch.zeros(nc, dtype=torch.int32, device=dev)
status[7] = 1
uf = torch.zeros((nc, 18 * 9 * 16), dtype=torch.float64, device=dev); wf = torch.zeros((nc, 13 * 9 * 64), dtype=torch.float64, device=dev)
ok = torch.zeros(nc, dtype=torch.int32, device=dev)
def run():
    _ffi.check(L.sf_debug_lowrank(P(lam_d), P(nuse), P(status), P(al_d), na, p, nc, P(uf), P(wf), P(ok), _ffi.stream_ptr()), "lowrank")
run(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): run()
b.record(); torch.cuda.synchronize()
okh = ok.cpu().numpy()
print("k_lowrank on %d columns: %.3f ms per launch; verdicts rank24/rank28/rank36/full = %d/%d/%d/%d" % (nc, a.elapsed_time(b) / 20, (okh == 3).sum(), (okh == 1).sum(), (okh == 2).sum(), (okh == 0).sum()))
use = okh > 0
np.savez(sys.argv[1], ufrag=uf.cpu().numpy()[use], wfrag=wf.cpu().numpy()[use], lrok=okh)
