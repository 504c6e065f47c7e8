#!/usr/bin/env python3
"""The score kernel alone on the full-size cube: `score_probe.py VARIANT [LPW]` runs it 5 times and prints the HIP-event
time; run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) for the HBM traffic of a variant."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from srcfinder_amd import _ffi
from srcfinder_amd.synth import make_cube_torch

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lpw = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lines, samples, p = 20000, int(os.environ.get('SF_PROBE_SAMPLES', '598')), 72
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
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
P = _ffi.ptr
L.sf_debug_set(1, variant); L.sf_debug_set(2, lpw)
def run():
    _ffi.check(L.sf_cmf_score(P(cube), lines, 425, samples, 0, samples, 350, p, P(filt), P(bias), P(status), P(aidx), P(nuse),
                              60, 42, 24, -9999.0, P(out), samples, 0, 4, None, None, P(ws), _ffi.stream_ptr()), "score")
run(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): run()
b.record(); torch.cuda.synchronize()
print("samples %d " % samples + "variant %d lpw %d: %.4f ms per launch, checksum %.6e" % (variant, lpw, a.elapsed_time(b) / 5, float(out[..., 3].sum())), "| %.3f ns per kilo-pixel" % (a.elapsed_time(b) / 5 * 1e6 / (lines * samples / 1e3)))
