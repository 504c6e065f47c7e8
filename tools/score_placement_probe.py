#!/usr/bin/env python3
"""Does the score kernel's duration depend on WHERE its product buffer lies relative to the cube?  The kernel alone on the
full-size cube, the product buffer a view at different byte offsets into one large allocation (and a few separately
allocated ones): HIP-event time per launch, 10 launches each, interleaved rounds.  (profiles/r05_score_placement.md)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from srcfinder_amd import _ffi
from srcfinder_amd.synth import make_cube_torch

lines, samples, p = 20000, 598, 72
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
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
nbytes = lines * samples * 32
big = torch.empty(nbytes + (80 << 20), dtype=torch.uint8, device=dev)
P = _ffi.ptr
offsets = [0, 256, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 3 << 20, 5 << 20, 8 << 20, 16 << 20, 33 << 20, 64 << 20]
views = {("view +%d" % o): big[o:o + nbytes].view(torch.float64).view(lines, samples, 4) for o in offsets}
pads = []
if len(sys.argv) > 1 and sys.argv[1] == "empty":
    torch.cuda.empty_cache()        # the segments the cube generator's temporaries left in the caching allocator go back to the driver
for k in range(10):
    pads.append(torch.empty((7 + 13 * k) << 20, dtype=torch.uint8, device=dev))     # perturb the allocator between them
    views["own alloc %d" % k] = torch.empty((lines, samples, 4), dtype=torch.float64, device=dev)

def run(out):
    _ffi.check(L.sf_cmf_score(P(cube), lines, 425, samples, 0, samples, 350, p, P(filt), P(bias), P(status), P(aidx), P(nuse),
                              60, 42, 24, -9999.0, P(out), samples, 0, 4, None, None, P(ws), _ffi.stream_ptr()), "score")

res = {k: [] for k in views}
for rnd in range(3):
    for k, out in views.items():
        run(out); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): run(out)
        b.record(); torch.cuda.synchronize()
        res[k].append(a.elapsed_time(b) / 10)
fill = {}
for k, out in views.items():        # does a plain fill of the buffer see the same placement effect?
    flat = out.view(-1)
    flat.fill_(1.0); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): flat.fill_(2.0)
    b.record(); torch.cuda.synchronize()
    fill[k] = a.elapsed_time(b) / 10
print("cube at 0x%x (mod 2 MiB: 0x%x)" % (cube.data_ptr(), cube.data_ptr() % (2 << 20)))
for k, out in views.items():
    print("%-18s out at 0x%x  (out - cube) mod 1 GiB = 0x%08x   ms per launch: %s   fill %.4f ms" %
          (k, out.data_ptr(), (out.data_ptr() - cube.data_ptr()) % (1 << 30), " ".join("%.4f" % v for v in res[k]), fill[k]))
