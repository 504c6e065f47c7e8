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
print("fuzz cnn: %d mismatches" % bad)
sys.exit(1 if bad else 0)
