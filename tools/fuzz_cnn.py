#!/usr/bin/env python3
"""Geometry sweep of the CNN paths against the torch-CPU oracle: FCN shift-and-stitch on planes whose sides are below,
at and above multiples of 32 (the divisibility pad is a full extra block when the side IS a multiple), 1 x 1 planes,
NODATA pixels; and the tile scorer on tiny planes.  python tools/fuzz_cnn.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
torch.set_num_threads(8)
from srcfinder_amd import cnn
from srcfinder_amd.cnn_weights import synthetic_plane, synthetic_state_dict
from oracle import cnn_oracle as O

sd = synthetic_state_dict(seed=2024)
net = cnn.GoogLeNetHIP(sd)
mean, std = O.MODEL_NORM["COVID_QC"]
rng = np.random.default_rng(0)

def logit(p):
    p = np.clip(p.astype(np.float64), 1e-300, 1.0)
    return np.log(p) - np.log1p(-np.minimum(p, 1.0 - 1e-16))

bad = 0
t0 = time.time()
for (H, W) in [(1, 1), (32, 64), (31, 33), (64, 32), (33, 95), (96, 31), (7, 130)]:
    plane = synthetic_plane(H, W, seed=int(rng.integers(1 << 30)))
    for _ in range(3):
        plane[int(rng.integers(H)), int(rng.integers(W))] = -9999.0
    got = cnn.fcn_predict_flightline(plane, "COVID_QC", net=net, batch=16, to_numpy=True)
    want, _ = O.fcn_predict_plane(plane, sd, mean, std)
    ok = np.array_equal(got == -9999, want == -9999)
    v = want != -9999
    ok = ok and np.allclose(got[v], want[v], rtol=5e-3, atol=2e-6)
    mid = v & (want > 1e-3) & (want < 1 - 1e-3)      # a float32 probability within 1e-3 of 0 or 1 no longer carries its logit
    if mid.any():
        ok = ok and np.abs(logit(got[mid]) - logit(want[mid])).max() < 5e-3
    print("fcn %3d x %3d: %s (%.0f s)" % (H, W, "ok" if ok else "MISMATCH", time.time() - t0), flush=True)
    bad += 0 if ok else 1
for (H, W) in [(1, 1), (2, 7), (5, 3)]:
    plane = synthetic_plane(H, W, seed=int(rng.integers(1 << 30)))
    if H * W > 2:
        plane[H - 1, W - 1] = -9999.0
    got = cnn.predict_flightline(plane, "COVID_QC", net=net, batch=8, to_numpy=True)
    want = O.predict_plane(plane, sd, mean, std)
    ok = np.array_equal(got == -9999, want == -9999) and np.allclose(got[want != -9999], want[want != -9999], rtol=2e-4, atol=1e-7)
    print("tiles %d x %d: %s (%.0f s)" % (H, W, "ok" if ok else "MISMATCH", time.time() - t0), flush=True)
    bad += 0 if ok else 1
# round 6: the shared trunk (route "split": phase maps + per-window rings, csrc/cnn_share.hip) against every window on its own, on
# random plane shapes (narrower / wider / taller than a window, sides at and around multiples of 4 and 8: every phase, map edges on
# both sides), batch sizes that straddle image rows, and row ranges -- the same bits as the unshared route, whatever batch and rows
for it in range(14):
    H, W = int(rng.integers(1, 420)), int(rng.integers(1, 330))
    if it == 0: H, W = 300, 8
    if it == 1: H, W = 9, 600
    batch = int(rng.choice([7, 33, 96, 256, 700]))
    r0 = int(rng.integers(0, H)); r1 = min(H, r0 + int(rng.integers(1, 4)))
    plane = synthetic_plane(H, W, seed=int(rng.integers(1 << 30)))
    plane[int(rng.integers(H)), int(rng.integers(W))] = -9999.0
    a = cnn.predict_flightline(plane, "COVID_QC", net=net, batch=batch, rows=(r0, r1), route="split")
    b = cnn.predict_flightline(plane, "COVID_QC", net=net, batch=batch, rows=(r0, r1), route="split_unshared")
    c = cnn.predict_flightline(plane, "COVID_QC", net=net, batch=max(1, batch // 2 + 1), rows=(r0, min(H, r1 + 1)), route="split")
    sa, sb = a[r0:r1], b[r0:r1]
    v = sb != -9999
    rel = float(((sa[v] - sb[v]).abs() / sb[v].abs().clamp_min(1e-7)).max()) if bool(v.any()) else 0.0
    ok = bool(torch.equal(sa, sb)) and bool(torch.equal(c[r0:r1], sa))      # bit-identical to every window on its own, and to itself
    print("shared trunk %3d x %3d batch %3d rows %d..%d: %s (max rel %.1e; %.0f s)" % (H, W, batch, r0, r1, "ok" if ok else "MISMATCH", rel,
                                                                                     time.time() - t0), flush=True)
    bad += 0 if ok else 1
# round 6, second half: longer row ranges -- strip-map groups of 16 image rows are rebuilt inside the call, batches straddle the group
# boundaries, and the range is scored as two concurrent halves (cnn.LANES) -- against one stream and against the unshared route
for it in range(6):
    H, W = int(rng.integers(40, 200)), int(rng.integers(20, 160))
    batch = int(rng.choice([33, 96, 256, 700]))
    r0 = int(rng.integers(0, H - 20)); r1 = min(H, r0 + int(rng.integers(17, 70)))
    plane = synthetic_plane(H, W, seed=int(rng.integers(1 << 30)))
    plane[int(rng.integers(H)), int(rng.integers(W))] = -9999.0
    a = cnn.predict_flightline(plane, "COVID_QC", net=net, batch=batch, rows=(r0, r1), route="split")
    b = cnn.predict_flightline(plane, "COVID_QC", net=net, batch=batch, rows=(r0, r1), route="split_unshared", lanes=1)
    c = cnn.predict_flightline(plane, "COVID_QC", net=net, batch=batch, rows=(r0, r1), route="split", lanes=1)
    ok = bool(torch.equal(a, b)) and bool(torch.equal(a, c))
    print("band sharing %3d x %3d batch %3d rows %d..%d: %s (%.0f s)" % (H, W, batch, r0, r1, "ok" if ok else "MISMATCH", time.time() - t0), flush=True)
    bad += 0 if ok else 1
print("fuzz cnn: %d mismatches" % bad)
sys.exit(1 if bad else 0)
