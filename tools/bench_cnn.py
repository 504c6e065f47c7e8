#!/usr/bin/env python3
"""Secondary benchmark: CNN tile scorer throughput (tiles/s, TFLOP/s) on one MI355X.
   python tools/bench_cnn.py --tiles 4096 --batch 256 [--cpu]
3.706 GFLOP per 256x256 tile (1.853 GMAC).  --cpu adds the torch-CPU oracle on a few tiles."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--no-fuse", action="store_true", help="conv1 and maxpool1 as two kernels (A/B of the fused kernel)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16"])
    ap.add_argument("--knob", action="append", default=[], help="key=value for sf_debug_set (repeatable)")
    ap.add_argument("--lanes", type=int, default=None, help="concurrent row parts (default: cnn.LANES = 2; 1 for a per-launch profile)")
    ap.add_argument("--route", default="split", help="split (shared trunk, C driver) | split_unshared | winograd | direct")
    ap.add_argument("--width", type=int, default=598, help="image width (whole rows are scored)")
    args = ap.parse_args()
    import torch
    from srcfinder_amd import cnn
    from srcfinder_amd.cnn_weights import synthetic_plane, synthetic_state_dict

    from srcfinder_amd import _ffi
    for kv in args.knob:
        k, v = kv.split("=")
        if _ffi.lib().sf_debug_set(int(k), int(v)) != 0:
            raise SystemExit("--knob %s: sf_debug_set refused the key" % kv)
    sd = synthetic_state_dict(2024)
    net = cnn.GoogLeNetHIP(sd, precision=args.precision)
    net.fuse_conv1 = not args.no_fuse
    w = args.width
    h = (args.tiles + w - 1) // w
    args.tiles = h * w                              # whole image rows: the C-side driver (sf_cnn_score_rows) sequences the graph
    plane = synthetic_plane(h, w, seed=5)
    ds = cnn.FlightlineConvolve(plane, "COVID_QC")
    out = torch.zeros(h * w, dtype=torch.float32, device="cuda")
    if args.precision == "fp32":
        net.calibrate(ds, args.batch)

    def run():
        cnn.score_tiles(net, ds, 0, args.tiles, args.batch, out, route=args.route if args.precision == "fp32" else None, lanes=args.lanes)

    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    line = {"metric": "CNN tiles/s (GoogLeNet 256x256 window per pixel, %s)" % args.precision, "value": round(args.tiles / dt, 1),
            "unit": "tiles/s", "tflops": round(args.tiles * 3.706e9 / dt / 1e12, 2), "batch": args.batch,
            "tiles": args.tiles, "route": args.route, "dtype": "f32" if args.precision == "fp32" else "f16 (fp32 accumulate)",
            "mfma_peak_tflops": 157.3 if args.precision == "fp32" else 2500.0}
    if args.cpu:
        from oracle import cnn_oracle as O
        torch.set_num_threads(os.cpu_count() or 1)
        n = 32
        t0 = time.perf_counter()
        O.predict_plane(plane, sd, *cnn.MODEL_NORM["COVID_QC"], batch=16, indices=range(n))
        t = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": round(n / t, 2), "unit": "tiles/s", "cores": os.cpu_count(), "kind": "port",
                                "sample": "%d tiles, torch CPU" % n}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
