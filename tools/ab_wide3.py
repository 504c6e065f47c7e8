#!/usr/bin/env python3
"""Full-band window (p = 425) on the benchmark flightline with several flightlines in flight: ms per flightline at depth 1 / 2 / 3.
    python tools/ab_wide3.py [key=value knobs ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from srcfinder_amd import cmf, _ffi
from srcfinder_amd.synth import make_cube_torch
from srcfinder_amd.inflight import FlightlinePipeline
lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
for kv in sys.argv[1:]:
    k, v = kv.split("="); _ffi.lib().sf_debug_set(int(k), int(v))
ns = 598
cube = make_cube_torch(20000, ns, seed=1234, abscf_full=lib[:, 2], nodata_column=ns // 3)
for depth in (1, 2, 3):
    outs = [torch.empty((20000, ns, 4), dtype=torch.float64, device="cuda") for _ in range(depth)]
    with FlightlinePipeline(depth, cube.device) as pipe:
        for i in range(depth):
            pipe.submit(cube, lib, out=outs[i], out_column0=0, active=(1, 425))
        pipe.synchronize()
        n = 2 * depth
        t0 = time.perf_counter()
        for i in range(n):
            pipe.submit(cube, lib, out=outs[i % depth], out_column0=0, active=(1, 425))
        pipe.synchronize()
        dt = (time.perf_counter() - t0) / n
    print("depth %d: %.1f ms per flightline  (knobs %s)" % (depth, dt * 1e3, " ".join(sys.argv[1:]) or "-"))
    del outs
    cmf._Workspace._bufs.clear(); torch.cuda.empty_cache()
