#!/usr/bin/env python3
"""Extract-kernel variants (sf_debug_set key 6) on the benchmark flightline: sf_cmf_extract_columns alone, timed with
events, and the xt / mask / column-sum outputs compared with the production form.   python tools/tune_extract.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srcfinder_amd import _ffi, cmf
from srcfinder_amd.synth import make_cube_torch

lib = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "ch4_library.npz"))["library"]
lines, samples = 20000, 598
cube = make_cube_torch(lines, samples, seed=1234, abscf_full=lib[:, 2], device="cuda", nodata_column=199)
L = _ffi.lib()
ref = None
for var in (5, 3, 0, 5, 100):
    L.sf_debug_set(6, 0 if var == 100 else var); L.sf_debug_set(19, 1 if var == 100 else 0)   # 100 = four-line tiles with PLAIN stores (default: non-temporal)
    r = cmf.robust_mf(cube, lib)          # warm
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        r = cmf.robust_mf(cube, lib)
    b.record()
    torch.cuda.synchronize()
    key = (r.out.clone(), r.alphaidx.clone(), r.colstats.clone())
    same = "" if ref is None else " identical=%s" % all(torch.equal(x, y) or bool(((x == y) | (x.isnan() & y.isnan())).all()) for x, y in zip(key, ref))
    if ref is None:
        ref = key
    print("extract variant %d: %.3f ms per flightline (one in flight)%s" % (var, a.elapsed_time(b) / 5, same), flush=True)
L.sf_debug_set(6, 0); L.sf_debug_set(19, 0)

# three flightlines in flight (the bench's headline mode)
from srcfinder_amd.inflight import FlightlinePipeline
for var in (0, 100, 0, 100):
    L.sf_debug_set(6, 0); L.sf_debug_set(19, 1 if var == 100 else 0)
    pipe = FlightlinePipeline(3)
    outs = [torch.empty((lines, samples, 4), dtype=torch.float64, device="cuda") for _ in range(3)]
    for i in range(6):
        pipe.submit(cube, lib, out=outs[pipe.slot_of_next()])
    pipe.synchronize()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for i in range(12):
        pipe.submit(cube, lib, out=outs[pipe.slot_of_next()])
    pipe.synchronize()
    dt = (time.perf_counter() - t0) / 12 * 1e3
    pipe.close()
    print("extract variant %d, three in flight: %.3f ms per flightline" % (var, dt), flush=True)
L.sf_debug_set(6, 0)
