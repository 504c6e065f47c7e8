#!/usr/bin/env python3
"""A/B the score kernel on a full-size cube (interleaved rounds in ONE process).

    python tools/tune_score.py [samples] [quick]

Configurations are (variant, -, workgroups per CU, -[, experiment bits]): sf_debug_set keys 1, 12, 13.
variant 0 = the production kernel (64-sample blocks, staged stores), 100 = round 1 (records stored by the lanes),
10 / 11 = k_score_blk2 (128-sample blocks, two samples per lane) with plain / non-temporal loads.  Every configuration
must reproduce round 1's product bit for bit (same FMA order)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from srcfinder_amd import _ffi
from srcfinder_amd.synth import make_cube_torch

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 598
quick = len(sys.argv) > 2
lines, p = 20000, 72
lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
cube = make_cube_torch(lines, samples, seed=1, abscf_full=lib[:, 2])
L = _ffi.lib()
dev = cube.device
g = torch.Generator(device=dev); g.manual_seed(3)
filt = torch.randn((samples, p), dtype=torch.float64, device=dev, generator=g)
bias = torch.randn(samples, dtype=torch.float64, device=dev, generator=g)
status = torch.zeros(samples, dtype=torch.int32, device=dev)
aidx = torch.full((samples,), 130, dtype=torch.int32, device=dev)
nuse = torch.full((samples,), lines, dtype=torch.int32, device=dev)
out = torch.empty((lines, samples, 4), dtype=torch.float64, device=dev)
colstats = torch.empty((3, samples), dtype=torch.float64, device=dev)
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
P = _ffi.ptr


def run():
    _ffi.check(L.sf_cmf_score(P(cube), lines, 425, samples, 0, samples, 350, p, P(filt), P(bias), P(status), P(aidx), P(nuse),
                              60, 42, 24, -9999.0, P(out), samples, 0, 4, None, P(colstats), P(ws), _ffi.stream_ptr()), "score")


def setcfg(c):
    L.sf_debug_set(1, c[0]); L.sf_debug_set(12, c[2])
    L.sf_debug_set(13, c[4] if len(c) > 4 else 0)


cfgs = [(100, 0, 0, 0), (0, 0, 0, 0), (10, 0, 0, 0), (11, 0, 0, 0)]
if os.environ.get("SF_SCORE_EXP"):     # library built with EXTRA=-DSF_SCORE_EXPERIMENTS: where does k_score_blk2's time go?
    # exp bits: 1 no epilogue, 4 no arithmetic, 8 no record assembly / stores, 32 one contiguous run per wave,
    #           64 staging but no global stores, 256 / 512 / 768 nt / sc1 / sc0 sc1 stores
    cfgs += [(10, 0, 0, 0, e) for e in (1, 4, 8, 32, 64, 256, 512, 768)]
cfgs = list(dict.fromkeys(cfgs))
setcfg((100, 0, 0, 0)); run(); torch.cuda.synchronize(); ref = out.clone(); refcs = colstats.clone()
res = {c: [] for c in cfgs}
bad = set()
for rnd in range(4):
    for c in cfgs:
        setcfg(c)
        try:
            if rnd == 0:
                out.zero_(); run(); torch.cuda.synchronize()
                if not torch.equal(out, ref):
                    bad.add(c)
                if c[0] != 100 and not torch.allclose(colstats, refcs, rtol=1e-9, atol=0, equal_nan=True):
                    bad.add(c)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(); run(); b.record(); torch.cuda.synchronize()
            res[c].append(a.elapsed_time(b) / 2)     # includes the tiny transpose + colstats kernels
        except _ffi.SrcfinderError as e:
            res[c].append(float("nan")); bad.add(c); print("cfg", c, "failed:", e)
setcfg((0, 0, 0, 0))
for c in sorted(cfgs, key=lambda c: np.median(res[c])):
    med = np.median(res[c])
    print("variant %3d BG %2d wgs/CU %d G %d exp %d : median %.3f ms min %.3f  -> %.0f GB/s (332 B/px), frac(4p+8) %.3f %s"
          % (c[0], c[1], c[2], c[3], c[4] if len(c) > 4 else 0, med, min(res[c]), 332 * lines * samples / med / 1e6,
             296 * lines * samples / (med * 1e-3) / 8e12, "MISMATCH" if c in bad else "bit-identical"))
