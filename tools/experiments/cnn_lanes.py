#!/usr/bin/env python3
"""Throughput of predict_flightline over 224 rows x 598 columns with 1 .. 4 concurrent row parts (cnn.LANES).  Round 6."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from srcfinder_amd import cnn
from srcfinder_amd.cnn_weights import synthetic_plane, synthetic_state_dict

W, rows, batch = 598, 224, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
net = cnn.GoogLeNetHIP(synthetic_state_dict(2024))
plane = synthetic_plane(rows, W, seed=5)
ref = None
for lanes in (1, 2, 3, 4, 1, 2, 3, 4):
    cnn.predict_flightline(plane, "COVID_QC", net=net, batch=batch, lanes=lanes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = cnn.predict_flightline(plane, "COVID_QC", net=net, batch=batch, lanes=lanes)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ref = out if ref is None else ref
    print("lanes %d: %.1f windows/s, equal to one lane: %s" % (lanes, rows * W / dt, bool(torch.equal(out, ref))), flush=True)
