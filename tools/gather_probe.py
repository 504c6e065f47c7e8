#!/usr/bin/env python3
"""Cost of the pieces of the column gather on one GPU (pack, collective with world 1, assemble)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
lines, ncols, samples, world = 20000, 75, 598, 8
out = torch.randn((lines, ncols, 4), dtype=torch.float64, device="cuda")
def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
x = out[..., 3].transpose(0, 1)
send = torch.zeros((ncols, lines), dtype=torch.float64, device="cuda")
print("zeros            %.3f ms" % timeit(lambda: torch.zeros((ncols, lines), dtype=torch.float64, device="cuda")))
print("pack copy        %.3f ms" % timeit(lambda: send.copy_(x)))
recv = [torch.empty_like(send) for _ in range(world)]
full = torch.empty((lines, samples), dtype=torch.float64, device="cuda")
def assemble():
    for r in range(world):
        a, b = r * samples // world, (r + 1) * samples // world
        full[:, a:b].copy_(recv[r][:b - a].transpose(0, 1))
print("assemble 8 blocks %.3f ms" % timeit(assemble))
print("contig 12MB copy %.3f ms" % timeit(lambda: recv[0].copy_(send)))
