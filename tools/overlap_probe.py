#!/usr/bin/env python3
"""Do two half-flightlines on two HIP streams overlap (memory-bound extract/score under the MFMA-bound sweep)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from srcfinder_amd import cmf, _ffi
from srcfinder_amd.synth import make_cube_torch

lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
lines = 20000
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
widths = [(598 * (i + 1)) // G - (598 * i) // G for i in range(G)]
halves = [make_cube_torch(lines, w, seed=1 + i, abscf_full=lib[:, 2]) for i, w in enumerate(widths)]
outs = [torch.empty((lines, w, 4), dtype=torch.float64, device="cuda") for w in widths]
# one workspace per stream
_orig = cmf._Workspace.get.__func__
bufs = {}
def get(cls, nbytes, device):
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    if key not in bufs or bufs[key].numel() < nbytes:
        bufs[key] = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    return bufs[key]
cmf._Workspace.get = classmethod(get)
streams = [torch.cuda.Stream() for _ in range(G)]

def run(concurrent):
    for i in range(G):
        st = streams[i] if concurrent else streams[0]
        with torch.cuda.stream(st):
            cmf.robust_mf(halves[i], lib, out=outs[i])

for mode in (False, True, False, True):
    run(mode); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        run(mode)
    torch.cuda.synchronize()
    print("concurrent" if mode else "sequential", "%.3f ms per flightline in %d parts" % ((time.perf_counter() - t0) / 5 * 1e3, G)) if False else print("concurrent" if mode else "sequential", "%.3f ms per flightline" % ((time.perf_counter() - t0) / 5 * 1e3))
