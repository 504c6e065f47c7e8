#!/usr/bin/env python3
"""Wide-window sweep A/B: the forms of k_wsweep (sf_debug_set(24, v): 0 = k_wsweep8: eight waves, wave-private operand slices; 4 = four waves, one per SIMD; 2 = eight waves on shared chunks;
1 = 32-row tiles, two workgroups per CU) through sf_cmf_wide_stats: NLL curve, alpha index, ms per call, phase clocks per form.
    python tools/ab_wsweep.py [ncols=128] [rows=20000] [p=425] [reps=2] [variants=0,2,0,2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from srcfinder_amd import _ffi, cmf

ncols = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
p = int(sys.argv[3]) if len(sys.argv) > 3 else 425
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
variants = [int(v) for v in (sys.argv[5] if len(sys.argv) > 5 else "4,0,4,0").split(",")]
f64 = 0
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
import ctypes
res = {}
buf = (ctypes.c_ulonglong * 8)()
for variant in variants:
    L.sf_debug_set(24, variant)
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
    L.sf_debug_set(22, int(os.environ.get('WS_STAMP', '1'))); L.sf_debug_wsweep_stamps(None, 1)
    run(); torch.cuda.synchronize()
    L.sf_debug_wsweep_stamps(buf, 1); L.sf_debug_set(22, 0)
    v = list(buf); nt = max(v[0], 1)
    cur = dict(nll=nll.cpu().numpy(), aidx=aidx.cpu().numpy())
    line = "variant %d: %9.2f ms/call  status!=0: %d  tiles %d: cycles per tile Y %.0f  r + rows %.0f" % (
        variant, ms, int((status != 0).sum()), v[0], v[1] / nt, v[2] / nt)
    if v[3]:
        line += " (wait %.0f first half %.0f second half %.0f exchange %.0f; the wave's tile %.0f)" % (v[3] / nt, v[4] / nt, v[5] / nt, v[6] / nt, v[7] / nt)
    if variant in res:
        line += "   re-run bit-identical: %s" % np.array_equal(res[variant]["nll"], cur["nll"], equal_nan=True)
    elif res:
        a = res[variants[0]]
        fin = np.isfinite(a["nll"]) & np.isfinite(cur["nll"])
        line += "   vs variant %d: NLL finite pattern equal %s, max rel diff %.2e, alpha index equal %s" % (
            variants[0], np.array_equal(np.isfinite(a["nll"]), np.isfinite(cur["nll"])),
            np.max(np.abs(a["nll"][fin] - cur["nll"][fin]) / np.abs(a["nll"][fin])), np.array_equal(a["aidx"], cur["aidx"]))
    res.setdefault(variant, cur)
    print(line, flush=True)
L.sf_debug_set(24, 0)
