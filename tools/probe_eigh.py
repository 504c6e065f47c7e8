#!/usr/bin/env python3
"""Time the batched eigensolver alone for different batch sizes (latency vs throughput)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from srcfinder_amd import _ffi
from srcfinder_amd.synth import synth_columns
L = _ffi.lib()
p = 72
mats = []
for seed in range(8):
    x = synth_columns(20000, p, 100 + seed); x -= x.mean(0)
    mats.append(np.cov(x.T))
P = _ffi.ptr
import itertools
for lpp, nc in itertools.product((8, 4, 16), (1, 75, 598)):
    L.sf_debug_set(7, lpp)
    S = torch.as_tensor(np.stack([mats[i % 8] for i in range(nc)])).cuda()
    nuse = torch.full((nc,), 20000, dtype=torch.int32, device="cuda")
    d = torch.empty((nc, p), dtype=torch.float64, device="cuda"); lam = torch.empty_like(d)
    ev = torch.empty((nc, p, p), dtype=torch.float64, device="cuda")
    st = torch.empty(nc, dtype=torch.int32, device="cuda")
    ws = torch.empty(L.sf_cmf_workspace_bytes(64, p, nc, 201), dtype=torch.uint8, device="cuda")
    def run():
        _ffi.check(L.sf_cmf_eigh(P(S), P(nuse), p, nc, P(d), P(lam), P(ev), P(st), P(ws), _ffi.stream_ptr()), "eigh")
    run(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); run(); run(); b.record(); torch.cuda.synchronize()
    stride = 30 * (p - 1) * (p // 2) * 2   # doubles per column of the rotation log (EIG_MAXSWEEP * steps * pairs * 2)
    sw = ws.view(torch.float64)[: nc * stride : stride].cpu().numpy()
    print("lanes/pair %d ncols %5d : %.3f ms per call; rotating sweeps min/max %d/%d" % (lpp, nc, a.elapsed_time(b) / 3, sw.min(), sw.max()))
    if "--stamps" in sys.argv:      # library built with -DSF_EIGH_STAMPS: s_memtime ticks (100 MHz?) summed over the Jacobi phase of column 0
        v = ws.view(torch.float64)[2:16:2].cpu().numpy()
        names = ("sweep prologue", "operands landed", "dot reduced", "rotation parameters", "rotation + stores issued", "barrier", "whole Jacobi phase")
        print("    stamps (ticks, column 0): " + ", ".join("%s %d" % (n, x) for n, x in zip(names, v)))
