#!/usr/bin/env python3
"""Host time to ENQUEUE one flightline (robust_mf) and one gather (world 1), against their GPU time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.distributed as dist
from srcfinder_amd import cmf, dist as sd
from srcfinder_amd.synth import make_cube_torch
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 75
cube = make_cube_torch(20000, NS, seed=1, abscf_full=lib[:, 2])
out = torch.empty((20000, NS, 4), dtype=torch.float64, device="cuda")
for _ in range(3):
    cmf.robust_mf(cube, lib, out=out); h = sd.gather_columns(out[..., 3], NS, dst=0, async_op=True); h.wait()
torch.cuda.synchronize()
N = 50
t0 = time.perf_counter()
for _ in range(N): cmf.robust_mf(cube, lib, out=out)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("robust_mf: host enqueue %.3f ms, total %.3f ms per call" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
t0 = time.perf_counter()
hs = [sd.gather_columns(out[..., 3], NS, dst=0, async_op=True) for _ in range(N)]
t1 = time.perf_counter()
for h in hs: h.wait()
t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
print("gather: host enqueue %.3f ms, wait+assemble enqueue %.3f ms, total %.3f ms per call" % ((t1 - t0) / N * 1e3, (t2 - t1) / N * 1e3, (t3 - t0) / N * 1e3))
dist.destroy_process_group()
