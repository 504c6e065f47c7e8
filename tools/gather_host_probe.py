#!/usr/bin/env python3
"""What does the gather path of a sharded bench step cost on one GPU (world 1, nccl)?  Variants:
   plain   pipeline only                       events  the event choreography without a collective
   comm    pack + gather on a comm stream      inline  pack + gather on the slot's own stream (no cross-stream fence)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch, torch.distributed as dist
from srcfinder_amd import cmf, dist as sd
from srcfinder_amd.inflight import FlightlinePipeline
from srcfinder_amd.synth import make_cube_torch
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
lines, ncols = 20000, 75
cube = make_cube_torch(lines, ncols, seed=1, abscf_full=lib[:, 2], device=dev)
N = 60

def run(mode):
    pipe = FlightlinePipeline(3, dev)
    outs = [torch.empty((lines, ncols, 4), dtype=torch.float64, device=dev) for _ in range(3)]
    comm = torch.cuda.Stream(device=dev)
    pend = [None] * 3
    for i in range(N + 6):
        if i == 6:
            torch.cuda.synchronize(); T0 = time.perf_counter()
        slot = pipe.slot_of_next()
        if pend[slot] is not None and mode in ("comm", "events"):
            h, packed = pend[slot]
            pipe.streams[slot].wait_event(packed)
            if h is not None:
                with torch.cuda.stream(comm):
                    h.wait()
        t = pipe.submit(cube, lib, out=outs[slot], out_column0=0)
        if mode == "comm":
            with torch.cuda.stream(comm):
                t.wait(comm)
                h = sd.gather_columns(outs[slot][..., 3], ncols, dst=0, async_op=True)
                packed = torch.cuda.Event(); packed.record(comm)
            pend[slot] = (h, packed)
        elif mode == "events":
            with torch.cuda.stream(comm):
                t.wait(comm)
                packed = torch.cuda.Event(); packed.record(comm)
            pend[slot] = (None, packed)
        elif mode == "inline":
            with torch.cuda.stream(pipe.streams[slot]):
                sd.gather_columns(outs[slot][..., 3], ncols, dst=0, async_op=False)
    pipe.synchronize(); torch.cuda.synchronize()
    dt = (time.perf_counter() - T0) / N * 1e3
    pipe.close()
    return dt

for mode in ("plain", "events", "comm", "inline", "plain"):
    print("%-7s %.3f ms per step" % (mode, run(mode)), flush=True)
dist.destroy_process_group()
