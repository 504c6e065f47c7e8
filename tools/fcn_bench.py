#!/usr/bin/env python3
"""FCN shift-and-stitch on a full-size plane (598 x 20000): seconds per flightline, fp32 and fp16, against the tile
scorer's rate.  usage: fcn_bench.py [nshifts=128] [batch=8]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from srcfinder_amd import cnn
from srcfinder_amd.cnn_weights import synthetic_plane, synthetic_state_dict

nsh = int(sys.argv[1]) if len(sys.argv) > 1 else 128
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
H, W = 20000, 598
plane = torch.as_tensor(synthetic_plane(H, W, seed=5)).cuda()
sd = synthetic_state_dict(seed=2024)
for prec in ("fp32", "fp16"):
    net = cnn.GoogLeNetHIP(sd, precision=prec)
    cnn.fcn_predict_flightline(plane, "COVID_QC", net=net, batch=batch, shifts=(0, batch))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cnn.fcn_predict_flightline(plane, "COVID_QC", net=net, batch=batch, shifts=(0, nsh))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    full = dt / nsh * 1024
    # one shift = the trunk over (H + pad + 32) x (W + pad + 32) pixels; a 256 x 256 tile is 65536 pixels
    Hc, Wc = H + (32 - H % 32) + 32, W + (32 - W % 32) + 32
    teq = Hc * Wc / 65536.0
    print("%s: %d shifts in %.2f s -> %.1f s per flightline (1024 shifts), %.0f tile-equivalents/s, peak mem %.1f GB"
          % (prec, nsh, dt, full, nsh * teq / dt, torch.cuda.max_memory_allocated() / 2**30))
    del net
    torch.cuda.empty_cache()
