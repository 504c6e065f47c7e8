#!/usr/bin/env python3
"""Time the batched narrow-window eigensolver alone for different batch sizes (latency vs throughput): the sweeps behind the
tridiagonal preconditioner (round 6, csrc/cmf_eigh_pre.h; sf_debug_set(7, 2)) against the plain sweeps from the Cholesky factor
(the default), with the eigenpairs of both held against numpy.linalg.eigh of the same correlation matrices."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from srcfinder_amd import _ffi
from srcfinder_amd.synth import synth_columns
L = _ffi.lib()
P = _ffi.ptr
import itertools
for p in (72, 83):
    mats = []
    for seed in range(8):
        x = synth_columns(20000, p, 100 + seed); x -= x.mean(0)
        mats.append(np.cov(x.T))
    for knob, nc in itertools.product((2, 0), (1, 75, 598)):
        L.sf_debug_set(7, knob)
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
        p2 = p + (p & 1)
        stride = 30 * (p2 - 1) * (p2 // 2) * 2   # doubles per column of the rotation log (EIG_MAXSWEEP * steps * pairs * 2)
        sw = ws.view(torch.float64)[: nc * stride : stride].cpu().numpy()
        # accuracy on the first 8 columns: eigenvalues relative to numpy's, residual and orthogonality of the vectors
        worst = [0.0, 0.0, 0.0]
        for c in range(min(nc, 8)):
            dd = np.sqrt(np.diag(mats[c])); R = mats[c] / np.outer(dd, dd)
            w = np.linalg.eigvalsh(R)
            lg, vg = lam[c].cpu().numpy(), ev[c].cpu().numpy()          # rows = eigenvectors
            o = np.argsort(lg)
            worst[0] = max(worst[0], np.max(np.abs(lg[o] - w) / w))
            worst[1] = max(worst[1], np.abs(R @ vg.T - vg.T * lg).max())
            worst[2] = max(worst[2], np.abs(vg @ vg.T - np.eye(p)).max())
        print("p %d %s ncols %5d : %.3f ms per call; rotating sweeps min/max %d/%d; eigenvalue rel err %.1e, residual %.1e, orthogonality %.1e"
              % (p, "precond" if knob == 2 else "plain  ", nc, a.elapsed_time(b) / 3, sw.min(), sw.max(), *worst), flush=True)
        if "--stamps" in sys.argv and knob == 2 and nc == 1:   # library built with EXTRA=-DSF_EIGH_STAMPS: cycle counts of column 0
            v = ws.view(torch.float64)[2:36:2].cpu().numpy()
            names = ("sweep prologue", "operands landed", "dot reduced", "rotation parameters", "rotation + stores issued", "barrier",
                     "whole Jacobi phase", "pre: load + tridiagonalisation", "pre: bounds + bisection", "pre: twisted vectors",
                     "pre: reflectors back", "pre: Cholesky", "pre: W", "pre: Gram + F0", "pre: correction + F", "pre: orthogonality check")
            print("    stamps (cycles, column 0): " + "; ".join("%s %d" % (n, x) for n, x in zip(names, v)))
L.sf_debug_set(7, 0)
