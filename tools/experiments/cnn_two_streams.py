#!/usr/bin/env python3
"""Does the tile scorer gain from two batches in flight?  Two host threads, two networks, two HIP streams on ONE device, each scoring
half of the rows through the C driver (route split, shared trunk) -- against one thread scoring all of them.  Round 6 experiment."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from srcfinder_amd import cnn
from srcfinder_amd.cnn_weights import synthetic_plane, synthetic_state_dict

W, rows, batch = 598, 224, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sd = synthetic_state_dict(2024)
plane = synthetic_plane(rows, W, seed=5)
nets = [cnn.GoogLeNetHIP(sd) for _ in range(2)]
dss = [cnn.FlightlineConvolve(plane, "COVID_QC") for _ in range(2)]
for n, d in zip(nets, dss):
    n.calibrate(d, batch)
out = torch.zeros(rows * W, dtype=torch.float32, device="cuda")
streams = [torch.cuda.Stream() for _ in range(2)]


def part(i, r0, r1):
    with torch.cuda.stream(streams[i]):
        cnn.score_tiles(nets[i], dss[i], r0 * W, r1 * W, batch, out, route="split")
        streams[i].synchronize()


def one():
    part(0, 0, rows)


def two():
    ts = [threading.Thread(target=part, args=(i, i * rows // 2, (i + 1) * rows // 2)) for i in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()


for name, fn in (("one stream", one), ("two streams", two), ("one stream", one), ("two streams", two)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s: %.1f windows/s" % (name, rows * W / dt), flush=True)
