#!/usr/bin/env python3
"""One flightline captured into a HIP graph (torch.cuda.CUDAGraph around cmf.robust_mf) against the same call issued eagerly:
ms per flightline, and whether the replayed product is bit-identical.  usage: graph_probe.py [samples]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from srcfinder_amd import cmf
from srcfinder_amd.synth import make_cube_torch
lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
lines = 20000
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 598
cube = make_cube_torch(lines, NS, seed=1, abscf_full=lib[:, 2])
out_e = torch.empty((lines, NS, 4), dtype=torch.float64, device="cuda")
out_g = torch.empty_like(out_e)
def timeit(f, n=40):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
eager = lambda: cmf.robust_mf(cube, lib, out=out_e)
print("%d samples eager    %.3f ms per flightline" % (NS, timeit(eager)))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    cmf.robust_mf(cube, lib, out=out_g)       # warm: workspace, LDS attributes
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        cmf.robust_mf(cube, lib, out=out_g)
torch.cuda.synchronize()
out_g.zero_()
print("%d samples graph    %.3f ms per flightline" % (NS, timeit(g.replay)))
print("%d samples eager    %.3f ms per flightline" % (NS, timeit(eager)))
print("bit-identical product:", torch.equal(out_e.view(torch.int64), out_g.view(torch.int64)))
