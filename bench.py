#!/usr/bin/env python3
"""Headline benchmark: columnwise matched filter (CMF) Mpixels/s on a 598 x 20000 x 425 float32 BIL cube.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one full pass of the hot path (extract -> mean -> covariance -> eigh -> LOO sweep -> filter -> score)
over the whole flightline, cube resident in HBM, outputs left on the device.  With N > 1 the cross-track
columns are sharded contiguously over the ranks (each rank holds only its own column slice of the cube) and
the step ends with ONE RCCL gather of the score blocks to rank 0 -- total work is fixed: "strong" scaling.
Rank 0 prints one JSON line (see DESIGN.md §Measurement for every field).
"""
import os

os.environ.setdefault("OMP_NUM_THREADS", "1")   # the CPU baseline is a scalar port: tiny LAPACK calls, 1 thread
import argparse
import json
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LINES, BANDS, SAMPLES = 20000, 425, 598
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s float4 copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--lines", type=int, default=LINES)
    ap.add_argument("--samples", type=int, default=SAMPLES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-columns", type=int, default=24)    # ~15-20 s of one host core
    ap.add_argument("--active", type=str, default="", help="a0,a1 (1-based inclusive) override of the active window, e.g. 1,425")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="flightlines in flight per GPU (srcfinder_amd.inflight); 0 = auto: 1 for >= 400 columns per "
                         "rank (the score kernel then runs alone and its HIP-event time is its isolated duration), else 3")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from srcfinder_amd import _ffi, cmf
    from srcfinder_amd import dist as sd
    from srcfinder_amd.synth import make_cube_torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dist = world == 1 and os.environ.get("SF_BENCH_FORCE_DIST") == "1"   # exercise the gather path on one GPU
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if force_dist:
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
    lines, samples = args.lines, args.samples
    s0, s1 = sd.shard_columns(samples, world, rank)
    ncols = s1 - s0
    a0, a1 = cmf.active_window("ch4", False)
    if args.active:
        a0, a1 = (int(v) for v in args.active.split(","))
    p = a1 - a0 + 1
    global ACTIVE
    ACTIVE = (a0, a1)

    # synthetic flightline: each rank generates only its own column slice [lines, 425, ncols]
    cube = make_cube_torch(lines, ncols, seed=1234 + rank, abscf_full=lib[:, 2], device=dev,
                           nodata_column=(ncols // 3))
    torch.cuda.synchronize()

    # A production run works through a queue of flightlines: `depth` of them are in flight on this GPU, each on its own
    # HIP stream with its own scratch and product buffer (srcfinder_amd/inflight.py), so one flightline's latency-bound
    # stages (eigensolver, rank factorisation: one workgroup per column) run beside another's streaming stages; the
    # gather of flightline i (RCCL, its own stream) overlaps the compute of the following ones.  Every step is still
    # one complete pass over the whole flightline and all K of them finish inside the timed region.
    from srcfinder_amd.inflight import FlightlinePipeline
    depth = args.in_flight if args.in_flight > 0 else (1 if ncols >= 400 else 3)
    pipe = FlightlinePipeline(depth, dev)
    outs = [torch.empty((lines, ncols, 4), dtype=torch.float64, device=dev) for _ in range(depth)]
    pending = [None] * depth                     # per slot: (gather handle, event "the slot's product has been packed")
    comm = torch.cuda.Stream(device=dev)         # packs and feeds the collective; never the stream submit() waits on
    state = {"last": None}

    def step():
        slot = pipe.slot_of_next()
        if pending[slot] is not None:
            h, packed = pending[slot]
            pipe.streams[slot].wait_event(packed)    # flightline i - depth's scores have left this slot's product buffer
            with torch.cuda.stream(comm):
                h.wait()                             # its image, assembled on rank 0
            pending[slot] = None
        t = pipe.submit(cube, lib, out=outs[slot], out_column0=0, active=(a0, a1))
        if world > 1 or force_dist:
            # the single RCCL gather of the score image: the float64 CMF band of every
            # rank's block, 8 B/pixel; the RGB copy stays with the rank that read those columns
            with torch.cuda.stream(comm):
                t.wait(comm)                         # only the comm stream waits for the slot's compute
                h = sd.gather_columns(outs[slot][..., 3], samples, dst=0, async_op=True)
                packed = torch.cuda.Event()
                packed.record(comm)
            pending[slot] = (h, packed)
        state["last"] = t
        return t.result

    def drain():
        with torch.cuda.stream(comm):
            for i in range(depth):
                if pending[i] is not None:
                    pending[i][0].wait()
                    pending[i] = None
        comm.synchronize()
        pipe.synchronize()

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(depth):                       # setup, not warmup: every slot allocates its scratch once
        step()
    drain()
    for _ in range(args.warmup):
        step()
    drain()
    L = _ffi.lib()
    barrier()
    flush_c_stdio()                              # RCCL's version banner (C stdout, buffered) goes out now, not after the result
    L.sf_cmf_score_timing(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    t_enq = time.perf_counter() - t0             # host time to enqueue the K steps (diagnostic)
    drain()                                      # every gather completes inside the timed region
    barrier()
    dt = time.perf_counter() - t0
    tot_ms = _ffi.C.c_double(0.0)
    nlaunch = _ffi.C.c_int(0)
    L.sf_cmf_score_timing_read(_ffi.C.byref(tot_ms), _ffi.C.byref(nlaunch))
    L.sf_cmf_score_timing(0)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        mpix = lines * samples / (dt / args.steps) / 1e6
        score_ms = tot_ms.value / max(nlaunch.value, 1)
        # algorithmic bytes of the score kernel per launch: every active value once
        # (4p B), the three RGB values (12 B), one 32-byte BIP record [R,G,B,CMF] float64 per pixel
        bytes_per_pixel = 4 * p + 12 + 32
        alg_bytes = bytes_per_pixel * lines * ncols
        achieved = alg_bytes / (score_ms * 1e-3) / 1e9 if score_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.isfile(pmc) and (lines, samples, p, world) == (LINES, SAMPLES, 72, 1):
            # HBM bytes per k_score launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE, see file)
            traffic = json.load(open(pmc))["kernels"]["k_score"]["hbm_bytes_per_launch"]
        line = {
            "metric": "CMF Mpixels/s on 598x20000x425 cube",
            "value": round(mpix, 3), "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "AVIRIS-NG flightline %d samples x %d lines x %d bands float32 BIL, active "
                                   "window %d..%d (p=%d%s), 201-point LOO shrinkage sweep, unimodal"
                                   % (samples, lines, BANDS, a0, a1, p, ", CH4 radiance" if (a0, a1) == (351, 422) else ""),
                       "parallelism": "columns sharded over %d rank(s), one RCCL gather" % world,
                       "in_flight": "%d flightlines in flight per GPU (one HIP stream each)" % depth,
                       "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 3),
                       "output": "float64 BIP [lines, samples, (R,G,B,CMF)]"},
            "roofline": {"bound": "hbm", "kernel": "k_score<true>", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "bytes_per_pixel": bytes_per_pixel,
                         # the same launch priced with SURVEY.md §8(d)'s score-only figure (4p + 8 B/pixel), i.e. not
                         # counting the RGB bands this kernel also reads and the 24 B of them it writes per pixel
                         "frac_score_only_4p_plus_8": round((4 * p + 8) * lines * ncols / (score_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                         if score_ms > 0 else 0.0,
                         "avg_launch_ms": round(score_ms, 4), "launches": nlaunch.value,
                         "measured": ("the kernel alone on the device (one flightline in flight)" if depth == 1 else
                                      "in situ: kernels of the %d other flightlines in flight share the device, so this is "
                                      "a lower bound of the kernel's own rate (run --in-flight 1 for the isolated number)"
                                      % (depth - 1))},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cube, lib, lines, ncols, args.cpu_columns, res)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        print(json.dumps(line), flush=True)      # the last line of the job's output


ACTIVE = None


def flush_c_stdio():
    """RCCL prints a version banner to the C stdout of every rank; left in the buffer it would be written at exit, AFTER
    the JSON line.  Flush it early so that the result is the last line of the job's output."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def cpu_baseline(cube, lib, lines, ncols, ncpu_cols, res):
    """The oracle (faithful numpy restatement of cmf/robust_mf.py, 201x det+inv+GEMM per column) timed on ONE
    host core over a bounded sample of the same cube: `ncpu_cols` evenly spaced columns, all lines."""
    from oracle import cmf_oracle as O
    cols = [int(round(i * (ncols - 1) / max(ncpu_cols - 1, 1))) for i in range(ncpu_cols)]
    cols = sorted(set(c for c in cols if c != ncols // 3))        # skip the all-NODATA column (no work)
    host = cube[:, :, cols].cpu().numpy()
    t0 = time.perf_counter()
    o = O.robust_mf_oracle(host, lib, active=ACTIVE)
    t = time.perf_counter() - t0
    # parity spot check of the timed sample against the GPU result of the same columns
    got = res.out[:, cols, 3].cpu().numpy()
    ref = o["out"][..., 3]
    nod = ref == -9999.0
    ok = bool(np.array_equal(got == -9999.0, nod)) and bool(
        np.all(np.abs(got[~nod] - ref[~nod]) <= 1e-4 * np.abs(ref[~nod]) + 1e-9 * np.abs(ref[~nod]).max()))
    so = o["status"] == 0
    aidx_ok = bool(np.array_equal(res.alphaidx.cpu().numpy()[cols][so], o["alphaidx"][so]))
    return {"value": round(lines * len(cols) / t / 1e6, 5), "unit": "Mpixel/s", "cores": 1, "kind": "port",
            "sample": "%d evenly spaced columns x %d lines of the benchmark cube, %.1f s, OMP_NUM_THREADS=1"
                      % (len(cols), lines, t),
            "parity_on_sample": ok and aidx_ok}


if __name__ == "__main__":
    main()
