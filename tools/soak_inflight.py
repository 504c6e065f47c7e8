#!/usr/bin/env python3
"""Soak test of srcfinder_amd.inflight: N flightlines through a depth-D pipeline, every result compared bit for bit
with the sequential result of the same cube.  usage: soak_inflight.py [N=300] [D=3]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from srcfinder_amd import cmf
from srcfinder_amd.inflight import FlightlinePipeline
from srcfinder_amd.synth import make_cube_torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
D = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
shapes = [(3000, 150), (2500, 75), (4100, 299), (1800, 64)]
cubes = [make_cube_torch(l, s, seed=50 + i, abscf_full=lib[:, 2]) for i, (l, s) in enumerate(shapes)]
ref = [cmf.robust_mf(c, lib, metadata=True) for c in cubes]
torch.cuda.synchronize()
bad = 0
with FlightlinePipeline(depth=D) as pipe:
    tickets = []
    for i in range(N):
        k = (i * 7 + i // 5) % len(cubes)
        tickets.append((k, pipe.submit(cubes[k], lib, metadata=True)))
        if len(tickets) >= 2 * D:
            kk, t = tickets.pop(0)
            r = t.synchronize()
            ok = torch.equal(r.out, ref[kk].out) and torch.equal(r.bgmeta, ref[kk].bgmeta) and torch.equal(r.alphaidx, ref[kk].alphaidx)
            bad += 0 if ok else 1
    for kk, t in tickets:
        r = t.synchronize()
        ok = torch.equal(r.out, ref[kk].out) and torch.equal(r.bgmeta, ref[kk].bgmeta) and torch.equal(r.alphaidx, ref[kk].alphaidx)
        bad += 0 if ok else 1
print("soak: %d flightlines, depth %d, mismatches %d" % (N, D, bad))
sys.exit(1 if bad else 0)
