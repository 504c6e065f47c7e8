#!/usr/bin/env python3
"""Does the score kernel's time depend on where the product buffer sits relative to the cube?  The kernel is run with
the same data and the output placed at different byte offsets inside one large allocation (bimodal 0.75 / 0.82 ms runs
were observed between processes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from srcfinder_amd import _ffi
from srcfinder_amd.synth import make_cube_torch
lines, samples, p = 20000, 598, 72
lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
cube = make_cube_torch(lines, samples, seed=1, abscf_full=lib[:, 2])
L = _ffi.lib(); P = _ffi.ptr
dev = cube.device
g = torch.Generator(device=dev); g.manual_seed(3)
filt = torch.randn((samples, p), dtype=torch.float64, device=dev, generator=g)
bias = torch.randn(samples, dtype=torch.float64, device=dev, generator=g)
status = torch.zeros(samples, dtype=torch.int32, device=dev)
aidx = torch.full((samples,), 130, dtype=torch.int32, device=dev)
nuse = torch.full((samples,), lines, dtype=torch.int32, device=dev)
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
n = lines * samples * 4
big = torch.empty(n + (64 << 20) // 8, dtype=torch.float64, device=dev)
print("cube ptr %% 2MiB = %d KiB, big ptr %% 2MiB = %d KiB" % ((cube.data_ptr() % (2 << 20)) >> 10, (big.data_ptr() % (2 << 20)) >> 10))
def run(out):
    _ffi.check(L.sf_cmf_score(P(cube), lines, 425, samples, 0, samples, 350, p, P(filt), P(bias), P(status), P(aidx), P(nuse),
                              60, 42, 24, -9999.0, P(out), samples, 0, 4, None, None, P(ws), _ffi.stream_ptr()), "score")
for off_kib in (0, 4, 64, 256, 512, 1024, 2048, 3072, 4096, 8192, 16384, 32768):
    out = big[off_kib * 128: off_kib * 128 + n].view(lines, samples, 4)
    run(out); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): run(out)
    b.record(); torch.cuda.synchronize()
    print("out offset %6d KiB: %.4f ms" % (off_kib, a.elapsed_time(b) / 5))
