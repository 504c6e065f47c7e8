#!/usr/bin/env python3
"""Whole flightlines in flight on DEPTH HIP streams (inter-flightline pipelining): ms per flightline, sequential and pipelined.
usage: pipeline_probe.py [samples] [depth] [key=value ...]   (sf_debug_set knobs for this thread, e.g. 20=1 8=4)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from srcfinder_amd import cmf
from srcfinder_amd.synth import make_cube_torch

lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
lines = 20000
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 598
DEPTH = int(sys.argv[2]) if len(sys.argv) > 2 else 2
from srcfinder_amd import _ffi
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    _ffi.lib().sf_debug_set(int(k), int(v))
cube = make_cube_torch(lines, NS, seed=1, abscf_full=lib[:, 2])
outs = [torch.empty((lines, NS, 4), dtype=torch.float64, device="cuda") for _ in range(DEPTH)]
bufs = {}
def get(cls, nbytes, device):
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    if key not in bufs or bufs[key].numel() < nbytes:
        bufs[key] = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    return bufs[key]
cmf._Workspace.get = classmethod(get)
streams = [torch.cuda.Stream() for _ in range(DEPTH)]
def run(n, concurrent):
    for i in range(n):
        with torch.cuda.stream(streams[i % DEPTH] if concurrent else streams[0]):
            cmf.robust_mf(cube, lib, out=outs[i % DEPTH])
for mode in (False, True, False, True):
    run(DEPTH, mode); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(40, mode); torch.cuda.synchronize()
    print(NS, "pipelined " if mode else "sequential", "%.3f ms per flightline" % ((time.perf_counter() - t0) / 40 * 1e3))
