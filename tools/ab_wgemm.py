#!/usr/bin/env python3
"""Wide-window statistics A/B: round 3's route (k_center + three 16x16x4 GEMMs + k_nllrows; sf_debug_set(23, 1)) against the fused
4x4x4 kernels of cmf_wgemm.hip (default) through sf_cmf_wide_stats: covariance, eigenvalues, NLL curve, alpha index, ms per call.
    python tools/ab_wgemm.py [ncols=36] [rows=20000] [p=425] [reps=2] [f64=0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from srcfinder_amd import _ffi, cmf

ncols = int(sys.argv[1]) if len(sys.argv) > 1 else 36
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
p = int(sys.argv[3]) if len(sys.argv) > 3 else 425
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
f64 = int(sys.argv[5]) if len(sys.argv) > 5 else 0
L = _ffi.lib()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(11)
ps = (p + 3) // 4 * 4
base = 5.0 * torch.exp(-3.0 * torch.arange(p, device=dev) / max(p - 1, 1)) + 0.2
xt = torch.zeros((ncols, rows, ps), dtype=torch.float64 if f64 else torch.float32, device=dev)
for c in range(ncols):
    lm = torch.randn((5, p), generator=g, device=dev) * 0.1 * base
    x = base + torch.randn((rows, 5), generator=g, device=dev) @ lm + torch.randn((rows, p), generator=g, device=dev) * 0.01 * base
    xt[c, :, :p] = x
xt[:, :, p:] = float("nan")                      # the padding of a row must never be read as data
mask = torch.ones((ncols, rows), dtype=torch.uint8, device=dev)
mask[:, :7] = 0
mask[:, rows // 2] = 0
xt[:, rows // 2, 3] = float("nan")               # an invalid row may hold anything
alphas_np = cmf.alpha_grid(); nalpha = len(alphas_np)
al = torch.as_tensor(alphas_np, device=dev)
f64k = dict(dtype=torch.float64, device=dev)
nuse = torch.empty(ncols, dtype=torch.int32, device=dev)
mu = torch.empty((ncols, p), **f64k)
ws = torch.empty(L.sf_cmf_workspace_bytes(rows, p, ncols, nalpha), dtype=torch.uint8, device=dev)
P, st = _ffi.ptr, _ffi.stream_ptr()
_ffi.check(L.sf_cmf_column_mean(P(xt), f64, P(mask), rows, p, ncols, P(nuse), P(mu), P(ws), st), "mean")
res = {}
for name, variant in (("round 3 (16x16x4, unfused)", 1), ("fused 4x4x4 (r4)", 0), ("round 3 again", 1), ("fused again", 0)):
    L.sf_debug_set(23, variant)
    S = torch.empty((ncols, p, p), **f64k); d = torch.empty((ncols, p), **f64k); lam = torch.empty((ncols, p), **f64k)
    evec = torch.empty((ncols, p, p), **f64k); status = torch.empty(ncols, dtype=torch.int32, device=dev)
    nll = torch.empty((ncols, nalpha), **f64k); aidx = torch.empty(ncols, dtype=torch.int32, device=dev)
    def run():
        _ffi.check(L.sf_cmf_wide_stats(P(xt), f64, P(mask), P(nuse), P(nuse), P(mu), P(al), nalpha, rows, p, ncols, P(S), P(d),
                                       P(lam), P(evec), P(status), P(nll), P(aidx), P(ws), st), "wide_stats")
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    res[name] = dict(S=S.cpu().numpy(), lam=torch.sort(lam, dim=1).values.cpu().numpy(), nll=nll.cpu().numpy(), aidx=aidx.cpu().numpy(),
                     status=status.cpu().numpy())
    print("%-28s %9.2f ms/call   status!=0: %d  S symmetric: %s  alpha idx %s" % (name, ms, int((status != 0).sum()),
          bool(torch.equal(S, S.transpose(1, 2))), aidx[:6].tolist()))
L.sf_debug_set(23, 0)
import ctypes
buf = (ctypes.c_ulonglong * 4)()
L.sf_debug_set(22, 1); L.sf_debug_wsweep_stamps(None, 1)
run(); torch.cuda.synchronize()
L.sf_debug_wsweep_stamps(buf, 1); L.sf_debug_set(22, 0)
v = list(buf); nt = max(v[0], 1)
print("sweep tiles %d: cycles per tile: Y = X W %.0f   r = Z C + rows %.0f" % (v[0], v[1] / nt, v[2] / nt))
a, b = res["round 3 (16x16x4, unfused)"], res["fused 4x4x4 (r4)"]
print("covariance: max |dS| / max |S| = %.2e" % (np.abs(a["S"] - b["S"]).max() / np.abs(a["S"]).max()))
print("eigenvalues: max rel diff %.2e" % np.max(np.abs(a["lam"] - b["lam"]) / np.abs(a["lam"])))
fin = np.isfinite(a["nll"]) & np.isfinite(b["nll"])
print("NLL: inf/nan pattern equal %s, max rel diff %.2e, alpha index equal %s"
      % (np.array_equal(np.isfinite(a["nll"]), np.isfinite(b["nll"])), np.max(np.abs(a["nll"][fin] - b["nll"][fin]) / np.abs(a["nll"][fin])),
         np.array_equal(a["aidx"], b["aidx"])))
print("re-run bit-identical: %s" % np.array_equal(res["fused 4x4x4 (r4)"]["nll"], res["fused again"]["nll"], equal_nan=True))
