#!/usr/bin/env python3
"""Full-band window (p = 425) on the benchmark flightline, one flightline at a time, for each value of ONE debug knob:
    python tools/ab_wide_knob.py KEY V1,V2,... [other key=value knobs ...]     (ms per flightline, alpha indices against the first value)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from srcfinder_amd import cmf, _ffi
from srcfinder_amd.synth import make_cube_torch
lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
key = int(sys.argv[1]); values = [int(v) for v in sys.argv[2].split(",")]
L = _ffi.lib()
for kv in sys.argv[3:]:
    k, v = kv.split("="); L.sf_debug_set(int(k), int(v))
ns = 598
cube = make_cube_torch(20000, ns, seed=1234, abscf_full=lib[:, 2], nodata_column=ns // 3)
out = torch.empty((20000, ns, 4), dtype=torch.float64, device="cuda")
ref = None
for v in values:
    L.sf_debug_set(key, v)
    res = cmf.robust_mf(cube, lib, out=out, active=(1, 425)); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): res = cmf.robust_mf(cube, lib, out=out, active=(1, 425))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 2 * 1e3
    ai = res.alphaidx.cpu().numpy(); sc = out[:, :, 3].clone()
    if ref is None: ref = (ai, sc)
    same = bool(np.array_equal(ai, ref[0])); d = float((sc - ref[1]).abs().max())
    print("key %d = %d: %.1f ms per flightline   alpha indices equal %s, max |score diff| %.3g" % (key, v, ms, same, d), flush=True)
