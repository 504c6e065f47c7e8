#!/usr/bin/env python3
"""The tridiagonal preconditioner of the wide eigensolver (csrc/cmf_wtri.hip) on benchmark-like correlation matrices:
F F^T = R, the cosines between F's columns, the tridiagonal eigenvalues against numpy.
    python tools/check_wtri.py [p=425] [nb=8] [rows=20000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from srcfinder_amd import _ffi
p = int(sys.argv[1]) if len(sys.argv) > 1 else 425
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
L = _ffi.lib()
rng = np.random.default_rng(5)
Rs, Ls = [], []
for m in range(nb):
    b = 5.0 * np.exp(-3.0 * np.arange(p) / (p - 1)) + 0.2
    lm = rng.standard_normal((5, p)) * 0.1 * b
    x = b + rng.standard_normal((n, 5)) @ lm + rng.standard_normal((n, p)) * 0.01 * b
    x = np.float64(np.float32(x)); x -= x.mean(0)
    S = x.T @ x / (n - 1); d = np.sqrt(np.diag(S)); R = S / np.outer(d, d); R = 0.5 * (R + R.T)
    Rs.append(R); Ls.append(np.linalg.cholesky(R))
R = np.stack(Rs); Lc = np.stack([l.T.copy() for l in Ls])          # column-major L = row-major L^T
dev = torch.device("cuda:0")
Rt = torch.as_tensor(R, device=dev); Lt = torch.as_tensor(Lc, device=dev)
F = torch.empty((nb, p, p), dtype=torch.float64, device=dev); tl = torch.empty((nb, p), dtype=torch.float64, device=dev)
pf = torch.empty(nb, dtype=torch.int32, device=dev)
ws = torch.empty(L.sf_debug_wtri_scratch_bytes(p, nb), dtype=torch.uint8, device=dev)
P, st = _ffi.ptr, _ffi.stream_ptr()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _ffi.check(L.sf_debug_wtri(P(Rt), P(Lt), p, nb, P(F), P(tl), P(pf), P(ws), st), "wtri")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("p = %d, %d matrices: %.2f ms per call; flags %s" % (p, nb, dt * 1e3, pf.cpu().numpy().tolist()[:8]))
import ctypes
buf = (ctypes.c_ulonglong * 8)()
L.sf_debug_wtri_stamps(buf, 1)
_ffi.check(L.sf_debug_wtri(P(Rt), P(Lt), p, nb, P(F), P(tl), P(pf), P(ws), st), "wtri"); torch.cuda.synchronize()
L.sf_debug_wtri_stamps(buf, 1)
v = list(buf)
print("tridiagonalisation, workgroup 0: cycles in reflector %d  symv %d  corrections %d  trailing updates %d  (%d columns)" % (v[0], v[1], v[2], v[3], v[4]))
Fh = F.cpu().numpy(); tlh = tl.cpu().numpy()
for m in range(min(nb, 4)):
    Fm = Fh[m].T                                # column-major buffer -> matrix
    ref = np.linalg.eigvalsh(R[m])
    M = Fm.T @ Fm; nn = np.sqrt(np.diag(M)); C = M / np.outer(nn, nn) - np.eye(p)
    print("matrix %d: |F F^T - R| / |R| = %.2e   max |cos| between columns %.2e (> 1e-9: %d)   tridiagonal eigenvalues: max rel err %.2e   "
          "squared column norms vs eigvalsh %.2e" % (m, np.abs(Fm @ Fm.T - R[m]).max() / np.abs(R[m]).max(), np.abs(C).max(),
          (np.abs(C) > 1e-9).sum() // 2, np.max(np.abs(tlh[m] - ref) / ref), np.max(np.abs(np.sort(np.diag(M)) - ref) / ref)))
