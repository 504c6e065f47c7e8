#!/usr/bin/env python3
"""Wide-window eigensolver A/B: round 3's blocked Jacobi with scalar rotations (the default) against the
Gram-space / MFMA form (sf_debug_set(10, 4), cmf_wjac.hip) through sf_cmf_wide_stats on one batch of synthetic columns:
eigen-residual |R v - lam v| / |R|, orthogonality |V^T V - I|, eigenvalue / NLL agreement, alpha index, ms per call.
    python tools/ab_wjac.py [ncols=36] [rows=1536] [p=425] [reps=3]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from srcfinder_amd import _ffi, cmf

ncols = int(sys.argv[1]) if len(sys.argv) > 1 else 36
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
p = int(sys.argv[3]) if len(sys.argv) > 3 else 425
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
L = _ffi.lib()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(7)
ps = (p + 3) // 4 * 4
base = 5.0 * torch.exp(-3.0 * torch.arange(p, device=dev) / max(p - 1, 1)) + 0.2
xt = torch.zeros((ncols, rows, ps), dtype=torch.float32, device=dev)
for c in range(ncols):
    lm = torch.randn((5, p), generator=g, device=dev) * 0.1 * base
    x = base + torch.randn((rows, 5), generator=g, device=dev) @ lm + torch.randn((rows, p), generator=g, device=dev) * 0.01 * base
    xt[c, :, :p] = x
mask = torch.ones((ncols, rows), dtype=torch.uint8, device=dev)
mask[:, :3] = 0
alphas_np = cmf.alpha_grid(); nalpha = len(alphas_np)
al = torch.as_tensor(alphas_np, device=dev)
f64 = dict(dtype=torch.float64, device=dev)
nuse = torch.empty(ncols, dtype=torch.int32, device=dev)
mu = torch.empty((ncols, p), **f64)
ws = torch.empty(L.sf_cmf_workspace_bytes(rows, p, ncols, nalpha), dtype=torch.uint8, device=dev)
P, st = _ffi.ptr, _ffi.stream_ptr()
_ffi.check(L.sf_cmf_column_mean(P(xt), 0, P(mask), rows, p, ncols, P(nuse), P(mu), P(ws), st), "mean")
res = {}
import ctypes
def stamps(variant):
    buf = (ctypes.c_ulonglong * 8)()
    L.sf_debug_set(10, variant)
    L.sf_debug_set(22, 1)
    L.sf_debug_wjac_stamps(None, 1)
    run(); torch.cuda.synchronize()
    L.sf_debug_wjac_stamps(buf, 1)
    L.sf_debug_set(22, 0)
    v = list(buf)
    nv, nf = max(v[0], 1), max(v[0] - v[6], 1)
    print("   phase clocks, cycles per visit: visits %d (ended after the Gram test: %d)  load %.0f  gram %.0f  rotations %.0f  update %.0f  store %.0f"
          % (v[0], v[6], v[1] / nv, v[2] / nv, v[3] / nf, v[4] / nf, v[5] / nf))
for name, variant in (("scalar (r3)", 5), ("quad visits (r4)", 0), ("gram/mfma (r4)", 4), ("scalar (r3) again", 5), ("quad visits (r4) again", 0)):
    L.sf_debug_set(10, variant)
    S = torch.empty((ncols, p, p), **f64); d = torch.empty((ncols, p), **f64); lam = torch.empty((ncols, p), **f64)
    evec = torch.empty((ncols, p, p), **f64); status = torch.empty(ncols, dtype=torch.int32, device=dev)
    nll = torch.empty((ncols, nalpha), **f64); aidx = torch.empty(ncols, dtype=torch.int32, device=dev)
    def run():
        _ffi.check(L.sf_cmf_wide_stats(P(xt), 0, P(mask), P(nuse), P(nuse), P(mu), P(al), nalpha, rows, p, ncols, P(S), P(d),
                                       P(lam), P(evec), P(status), P(nll), P(aidx), P(ws), st), "wide_stats")
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    dd = d[:, :, None] * d[:, None, :]
    R = S / dd
    V = evec.transpose(1, 2)                      # columns = eigenvectors
    resid = (R @ V - V * lam[:, None, :]).abs().amax(dim=(1, 2)) / R.abs().amax(dim=(1, 2))
    orth = (V.transpose(1, 2) @ V - torch.eye(p, **f64)).abs().amax(dim=(1, 2))
    res[name] = dict(lam=torch.sort(lam, dim=1).values.cpu().numpy(), nll=nll.cpu().numpy(), aidx=aidx.cpu().numpy(),
                     status=status.cpu().numpy())
    if variant in (4,) and "again" not in name:
        stamps(variant)
    print("%-22s %8.2f ms/call   max residual %.2e   max |V^T V - I| %.2e   status!=0: %d   lam range %.2e .. %.2e"
          % (name, ms, float(resid.max()), float(orth.max()), int((status != 0).sum()), float(lam.min()), float(lam.max())))
L.sf_debug_set(10, 0)
a, b = res["scalar (r3)"], res["quad visits (r4)"]
print("eigenvalues: max rel diff %.2e" % np.max(np.abs(a["lam"] - b["lam"]) / np.abs(a["lam"])))
fin = np.isfinite(a["nll"]) & np.isfinite(b["nll"])
print("NLL: inf pattern equal %s, max rel diff %.2e, alpha index equal %s (%s)"
      % (np.array_equal(np.isfinite(a["nll"]), np.isfinite(b["nll"])), np.max(np.abs(a["nll"][fin] - b["nll"][fin]) / np.abs(a["nll"][fin])),
         np.array_equal(a["aidx"], b["aidx"]), a["aidx"][:8]))
