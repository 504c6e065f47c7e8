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
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--lines", type=int, default=LINES)
    ap.add_argument("--samples", type=int, default=SAMPLES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cnn", action="store_true", help="skip the CNN tile-scorer section of the line (BASELINE configs 4/5)")
    ap.add_argument("--cnn-tiles", type=int, default=16384)
    ap.add_argument("--cnn-batch", type=int, default=1024)
    ap.add_argument("--no-wide", action="store_true", help="skip the full-band (p = 425) section of the line (SURVEY 8(d) config F425)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the CMF -> CNN end-to-end section (BASELINE config 4)")
    ap.add_argument("--no-ceiling", action="store_true",
                    help="skip the pure-traffic microbenchmark child process (always skipped under a profiler)")
    ap.add_argument("--no-ingest", action="store_true", help="skip the file -> HBM -> product section (SURVEY 8(f) N3)")
    ap.add_argument("--no-routes", action="store_true", help="skip the sweep-route histogram and the hard-spectrum line")
    ap.add_argument("--no-windows", action="store_true",
                    help="skip the CO2 (p = 83) and reflectance (-R, p = 416) window sections (cmf/robust_mf.py:186-191)")
    ap.add_argument("--cpu-columns", type=int, default=12)    # ~7 s of one host core (+ the all-cores sample)
    ap.add_argument("--active", type=str, default="", help="a0,a1 (1-based inclusive) override of the active window, e.g. 1,425")
    ap.add_argument("--no-placement", action="store_true", help="skip the product-buffer placement pick of the setup phase")
    ap.add_argument("--knob", action="append", default=[], help="key=value for sf_debug_set (tuning / A-B runs; repeatable)")
    ap.add_argument("--replicas", action="store_true",
                    help="BASELINE config 5 instead of the sharded headline: every rank runs a WHOLE flightline (its own synthetic cube) "
                         "through CMF + CNN end to end; no collective on the data path; per-GPU and aggregate throughput")
    ap.add_argument("--strip-lines", type=int, default=2500, help="lines of the CMF plane the tile scorer measures (e2e / --replicas)")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="flightlines in flight per GPU (srcfinder_amd.inflight); 0 = the default, 3 for every N")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as a CHILD torch.distributed.run, before
        # this process has imported torch or touched the GPU (never an exec of a process that initialised HIP)
        raise SystemExit(launch_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    from srcfinder_amd import _ffi, cmf
    from srcfinder_amd import dist as sd
    from srcfinder_amd.synth import make_cube_torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one rank per GPU (torch.distributed.run --nproc-per-node %d), "
                         "or run `python bench.py --gpus %d` with no WORLD_SIZE in the environment and it starts them itself"
                         % (args.gpus, world, args.gpus, args.gpus))
    # SF_BENCH_SHARE_GPU=1 + SF_BENCH_BACKEND=gloo: N ranks on ONE GPU with host-staged collectives (RCCL refuses two ranks on
    # one device).  A FUNCTIONAL run of the N > 1 path on hardware where only one GPU can be leased -- the ranks share the
    # chip, so its value says nothing about scaling (profiles/r05_two_ranks_one_gpu.md)
    share_gpu = os.environ.get("SF_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("SF_BENCH_BACKEND", "nccl")
    if share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dist = world == 1 and os.environ.get("SF_BENCH_FORCE_DIST") == "1"   # exercise the gather path on one GPU
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if force_dist:
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
    lines, samples = args.lines, args.samples
    if args.replicas:
        return replicas_main(args, lib, rank, world, dev, backend, share_gpu, force_dist)
    s0, s1 = sd.shard_columns(samples, world, rank)
    ncols = s1 - s0
    a0, a1 = cmf.active_window("ch4", False)
    if args.active:
        a0, a1 = (int(v) for v in args.active.split(","))
    p = a1 - a0 + 1

    # synthetic flightline: each rank generates only its own column slice [lines, 425, ncols]
    cube = make_cube_torch(lines, ncols, seed=1234 + rank, abscf_full=lib[:, 2], device=dev,
                           nodata_column=(ncols // 3))
    torch.cuda.synchronize()
    # The generator's temporaries (GB-sized blocks) stay in torch's caching allocator, and product buffers carved out of those
    # segments make the HBM-bound kernels up to 8 % slower than buffers in fresh segments (score kernel 0.807 against 0.735-0.75 ms
    # for the same launch, persistent per buffer: tools/score_placement_probe.py, profiles/r05_score_placement.md).  Data
    # generation is not part of the path: its leftovers go back to the driver before the pipeline allocates anything.
    torch.cuda.empty_cache()

    # A production run works through a queue of flightlines: `depth` of them are in flight on this GPU, each on its own
    # HIP stream with its own scratch and product buffer (srcfinder_amd/inflight.py), so one flightline's latency-bound
    # stages (eigensolver, rank factorisation: one workgroup per column) run beside another's streaming stages; the
    # gather of flightline i (RCCL, behind it on its slot's stream) overlaps the compute of the other slots.  Every step is still
    # one complete pass over the whole flightline and all K of them finish inside the timed region.
    # ONE depth policy for every N (default 3): the 1 / 2 / 4 / 8-GPU values are then measured the same way.  The same K
    # steps are then repeated with ONE flightline in flight: that pass gives the latency of a single flightline
    # ("one_in_flight") and the score kernel's isolated duration for the roofline (with several flightlines in flight
    # its HIP-event time includes kernels of the other streams).
    from srcfinder_amd.inflight import FlightlinePipeline
    L = _ffi.lib()
    for kv in args.knob:
        k, v = kv.split("=")
        if L.sf_debug_set(int(k), int(v)) != 0:
            raise SystemExit("--knob %s: sf_debug_set refused the key (a retired or unknown knob: the A/B run would measure "
                             "the same code twice)" % kv)

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_pass(depth):
        torch.cuda.empty_cache()                 # (placement, see above: no product buffer out of an earlier pass's freed blocks)
        pipe = FlightlinePipeline(depth, dev)
        if not args.no_placement:
            # the library's own opt-in buffer pool (FlightlinePipeline.pick_product_buffers): what a long-running
            # host does once for its product buffers -- set-up, untimed; --no-placement = plain torch.empty buffers
            outs, placement = pipe.pick_product_buffers(cube, lib, spares=3, shape=(lines, ncols, 4), active=(a0, a1))
        else:
            outs = [torch.empty((lines, ncols, 4), dtype=torch.float64, device=dev) for _ in range(depth)]
            placement = None
        pending = [None] * depth                 # per slot: the gather handle of the flightline that used it last
        state = {}

        def step():
            slot = pipe.slot_of_next()
            st = pipe.streams[slot]
            if pending[slot] is not None:
                with torch.cuda.stream(st):
                    pending[slot].wait()                 # flightline i - depth: its image, assembled on rank 0
                pending[slot] = None
            t = pipe.submit(cube, lib, out=outs[slot], out_column0=0, active=(a0, a1))
            if world > 1 or force_dist:
                # the single RCCL gather of the score image: the float64 CMF band of every rank's block,
                # 8 B/pixel; the RGB copy stays with the rank that read those columns.  Enqueued on the SLOT's own stream,
                # behind its flightline: the other slots' flightlines run beside it, and no stream waits on another one's
                # event -- a cross-stream fence per step (the first version packed and gathered on a communication stream)
                # costs 0.65 ms of a 1.65 ms shard step on this stack (tools/gather_host_probe.py)
                with torch.cuda.stream(st):
                    pending[slot] = sd.gather_columns(outs[slot][..., 3], samples, dst=0, async_op=True)
            state["res"] = t.result

        def drain():
            for i in range(depth):
                if pending[i] is not None:
                    with torch.cuda.stream(pipe.streams[i]):
                        state["gathered"] = pending[i].wait()      # rank 0: the assembled [lines, samples] score image
                    pending[i] = None
            pipe.synchronize()

        for _ in range(depth):                   # setup, not warmup: every slot allocates its scratch once
            step()
        drain()
        for _ in range(args.warmup):
            step()
        drain()
        barrier()
        flush_c_stdio()                          # RCCL's version banner (C stdout, buffered) goes out now, not after the result
        L.sf_cmf_score_timing(1)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        t_enq = time.perf_counter() - t0         # host time to enqueue the K steps (diagnostic)
        drain()                                  # every gather completes inside the timed region
        barrier()
        dt = time.perf_counter() - t0
        tot_ms = _ffi.C.c_double(0.0)
        nlaunch = _ffi.C.c_int(0)
        L.sf_cmf_score_timing_read(_ffi.C.byref(tot_ms), _ffi.C.byref(nlaunch))
        L.sf_cmf_score_timing(0)
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        res = state["res"]
        # N > 1 (outside the timed region): every rank's block of the last flightline against its column range of the image
        # rank 0 assembled -- number of NODATA pixels exact, sum of the scores to rounding
        verified = None
        if world > 1 or force_dist:
            blk = outs[0][..., 3]
            ok = blk != -9999.0
            mine = torch.stack([blk[ok].sum(), ok.sum().to(torch.float64)]).to(dev if backend == "nccl" else "cpu")
            every = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            if rank == 0:
                img = state["gathered"]
                verified = tuple(img.shape) == (lines, samples)
                for r in range(world):
                    c0, c1 = sd.shard_columns(samples, world, r)
                    part = img[:, c0:c1]
                    okr = part != -9999.0
                    want = every[r].to(part.device)
                    verified = verified and float(okr.sum()) == float(want[1]) and \
                        abs(float(part[okr].sum()) - float(want[0])) <= 1e-9 * max(1.0, abs(float(want[0])))
        pipe.close()
        return {"dt": float(tmax.item()), "t_enq": t_enq, "score_ms": tot_ms.value / max(nlaunch.value, 1),
                "launches": nlaunch.value, "res": res, "outs": outs, "gather_verified": verified, "placement": placement}

    depth = args.in_flight if args.in_flight > 0 else 3
    main = timed_pass(depth)
    solo = main if depth == 1 else timed_pass(1)
    res = solo["res"]

    if rank == 0:
        ms_per_step = main["dt"] / args.steps * 1e3
        mpix = lines * samples / (main["dt"] / args.steps) / 1e6
        score_ms = solo["score_ms"]
        npix = lines * ncols
        # algorithmic bytes of the score kernel = every active value once (4p B) + one float64 score
        # (8 B) per pixel; every pixel counts (a NODATA row has to be read to be recognised and its record is written).
        # The kernel also copies the three RGB bands into the 32-byte BIP record (12 B read + 24 B written per pixel):
        # that is real traffic of this fused launch but NOT part of §8(d)'s figure -- reported as `frac_with_fused_rgb`.
        alg_bytes = (4 * p + 8) * npix
        achieved = alg_bytes / (score_ms * 1e-3) / 1e9 if score_ms > 0 else 0.0
        fused_bytes = (4 * p + 12 + 32) * npix
        line = {
            "metric": "CMF Mpixels/s on 598x20000x425 cube",
            "value": round(mpix, 3), "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "AVIRIS-NG flightline %d samples x %d lines x %d bands float32 BIL, active "
                                   "window %d..%d (p=%d%s), 201-point LOO shrinkage sweep, unimodal"
                                   % (samples, lines, BANDS, a0, a1, p, ", CH4 radiance" if (a0, a1) == (351, 422) else ""),
                       "parallelism": ("columns sharded over %d rank(s), one RCCL gather" % world) if not share_gpu else
                       ("FUNCTIONAL RUN: %d ranks sharing ONE GPU, host-staged %s gather -- not a scaling measurement" % (world, backend)),
                       "in_flight": "%d flightlines in flight per GPU (one HIP stream each); the same depth for every N"
                                    % depth,
                       "gather_verified": main["gather_verified"],
                       "product_buffer_placement": {"three_in_flight": main["placement"], "one_in_flight": solo["placement"],
                                                    "note": "FlightlinePipeline.pick_product_buffers (library API, set-up, untimed): score-kernel "
                                                            "ms of one flightline into each candidate product buffer; the fastest are "
                                                            "kept (profiles/r05_score_placement.md)"},
                       "host_enqueue_ms_per_step": round(main["t_enq"] / args.steps * 1e3, 3),
                       "one_in_flight": {"ms_per_step": round(solo["dt"] / args.steps * 1e3, 3),
                                         "value": round(lines * samples / (solo["dt"] / args.steps) / 1e6, 3),
                                         "note": "the same K steps with ONE flightline in flight (latency of a flightline)"},
                       "output": "float64 BIP [lines, samples, (R,G,B,CMF)]"},
            "roofline": {"bound": "hbm", "kernel": "k_score<RGB> (64-sample blocks, staged stores)", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": None, "bytes_per_pixel": 4 * p + 8,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "frac_with_fused_rgb": round(fused_bytes / (score_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                         if score_ms > 0 else 0.0,
                         "avg_launch_ms": round(score_ms, 4), "launches": solo["launches"],
                         "measured": "HIP events on the launch stream around the kernel alone, one flightline in flight"},
        }
        line["roofline"].update(pmc_traffic(lines, samples, p, world))
        if world == 1 and (lines, samples, p) == (LINES, SAMPLES, 72) and not args.no_ceiling and not under_profiler():
            line["roofline"].update(measured_ceiling(score_ms))
            line["roofline"].update(in_step_traffic(cube, lib, (a0, a1)))
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cube, lib, lines, ncols, args.cpu_columns, res, (a0, a1))
        if world == 1 and not args.no_cnn:
            # the second half of the path (BASELINE.json configs 4 / 5): the per-pixel GoogLeNet tile scorer on the CMF plane
            # this run produced -- its own metric (tiles/s) and its own roofline (fp32 MFMA), inside the same line
            line["cnn"] = cnn_section(res, args.cnn_tiles, args.cnn_batch, not args.no_cpu_baseline)
        full = world == 1 and (lines, samples, p) == (LINES, SAMPLES, 72)
        if full and not args.no_routes:
            # which sweep kernel each column took (the step's cost depends on it) and the same step on a hard spectrum
            line["sweep_routes"] = cmf.sweep_routes(cube, lib, active=(a0, a1))
            line["hard_spectrum"] = hard_spectrum_section(cube, lib, (a0, a1), depth)
        if full and not args.no_windows:
            # the reference's other two windows (cmf/robust_mf.py:186-191): CO2 radiance 309..391 (p = 83) at the headline's
            # depth, and the -R reflectance window 5..420 (p = 416, the wide route) one flightline at a time
            line["co2"] = window_section(cube, lib, cmf.active_window("co2", False), False, depth, 10, ms_per_step)
        if full and not args.no_ingest:
            line["ingest"] = ingest_section(cube, lib, res)
        if full and not args.no_e2e:
            # BASELINE config 4 made visible to the driver: cube -> CMF -> saliency map of the WHOLE flightline
            line["e2e"] = e2e_section(cube, lib, solo["dt"] / args.steps, line.get("cnn"))
        if full and not args.no_wide:
            # SURVEY 8(d) lists the full-band window among the configs: the batched-GEMM path, bounded to three flightlines
            del main, solo
            if not args.no_windows:
                line["reflectance"] = window_section(cube, lib, cmf.active_window("ch4", True), True, 1, 2, ms_per_step)
            line["wide"] = wide_section(cube, lib, with_cpu=not args.no_cpu_baseline)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        print(json.dumps(line), flush=True)      # the last line of the job's output


def replicas_main(args, lib, rank, world, dev, backend, share_gpu, force_dist):
    """BASELINE config 5: a batch of flightlines, one per GPU -- every rank takes a whole flightline (its own synthetic cube) through
    the CMF (K steps, three in flight, as the headline) and the tile scorer (a measured strip of --strip-lines full-width lines of
    its CMF plane, scaled to the flightline) and reports cube -> saliency map seconds.  Replicas only: no collective on the data
    path; the ranks' figures meet in one all_gather AFTER the timed regions.  value = the aggregate Mpixel/s of the
    flightlines in flight on the node; scaling "weak" (per-GPU work is fixed as N grows)."""
    import torch
    import torch.distributed as dist
    from srcfinder_amd import cmf
    from srcfinder_amd.inflight import FlightlinePipeline
    from srcfinder_amd.synth import make_cube_torch
    lines, samples = args.lines, args.samples
    cube = make_cube_torch(lines, samples, seed=1234 + rank, abscf_full=lib[:, 2], device=dev, nodata_column=samples // 3)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    depth = args.in_flight if args.in_flight > 0 else 3
    outs = [torch.empty((lines, samples, 4), dtype=torch.float64, device=dev) for _ in range(depth)]
    with FlightlinePipeline(depth, dev) as pipe:
        for i in range(depth + args.warmup):
            pipe.submit(cube, lib, out=outs[i % depth], out_column0=0)
        pipe.synchronize()
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            pipe.submit(cube, lib, out=outs[i % depth], out_column0=0)
        pipe.synchronize()
        cmf_s = (time.perf_counter() - t0) / args.steps
    del outs
    sec = e2e_section(cube, lib, cmf_s, None, strip_lines=min(args.strip_lines, lines), with_fcn=False)
    mine = torch.tensor([sec["seconds"], cmf_s, sec["strip"]["windows_per_s"]], dtype=torch.float64,
                        device=dev if backend == "nccl" else "cpu")
    every = [torch.empty_like(mine) for _ in range(world)]
    if world > 1 or force_dist:
        dist.all_gather(every, mine)
        dist.barrier()
        dist.destroy_process_group()
    else:
        every = [mine]
    flush_c_stdio()
    if rank == 0:
        per = [{"rank": r, "seconds_per_flightline": round(float(e[0]), 2), "cmf_ms": round(float(e[1]) * 1e3, 3),
                "cnn_windows_per_s": round(float(e[2]), 1), "value": round(lines * samples / float(e[0]) / 1e6, 4)} for r, e in enumerate(every)]
        slow = max(p["seconds_per_flightline"] for p in per)
        line = {"metric": "CMF + CNN saliency map end to end, one flightline per GPU (BASELINE config 5)",
                "value": round(sum(p["value"] for p in per), 4), "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(slow * 1e3, 1), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f64 (CMF) + f32 (CNN, split-operand route)", "data": "synthetic",
                "config": {"workload": "%d flightline(s) of %d samples x %d lines x %d bands, one per GPU: CMF (CH4 window, 201-point "
                                       "sweep) -> per-pixel 256 x 256 GoogLeNet tile scorer" % (world, samples, lines, BANDS),
                           "parallelism": ("replicas only: no collective on the data path" if not share_gpu else
                                           "FUNCTIONAL RUN: %d replicas sharing ONE GPU -- not a scaling measurement" % world),
                           "cnn_measured_on": "%d full-width lines of each rank's CMF plane, scaled to the flightline" % min(args.strip_lines, lines),
                           "ms_per_step_is": "seconds per flightline of the slowest replica, in ms"},
                "per_gpu": per}
        print(json.dumps(line), flush=True)


def launch_ranks(n):
    """`python bench.py --gpus N` (N > 1) with no launcher around it: run `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <the same arguments>` as a child process and relay
    its output.  Called before torch is imported, so this parent never initialises the GPU; the child is a child, not an exec.
    Rank 0's JSON line is printed again as the LAST line of this process's stdout (the ranks' stderr / banners may trail it in
    the child's stream); the return value is the child's exit code."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC only on this host driver (RCCL peers)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
    result = None
    for ln in proc.stdout:
        s = ln.strip()
        if s.startswith('{"metric"') and s.endswith("}"):
            result = s                                       # held back: printed last
        else:
            sys.stdout.write(ln)
            sys.stdout.flush()
    rc = proc.wait()
    if result is not None:
        print(result, flush=True)
    elif rc == 0:
        print("bench.py: the %d ranks exited 0 but printed no result line" % n, file=sys.stderr)
        rc = 1
    return rc


def window_section(cube, lib, active, reflectance, depth, steps, ch4_ms):
    """The same flightline through another of the reference's active windows (cmf/robust_mf.py:186-191): `steps` passes with
    `depth` flightlines in flight, after one untimed pass per slot.  `per_pixel_vs_ch4` = this step's time over the CH4
    headline step's (the same pixel count)."""
    import torch
    from srcfinder_amd import cmf
    from srcfinder_amd.inflight import FlightlinePipeline
    lines, bands, ncols = cube.shape
    a0, a1 = active
    p = a1 - a0 + 1
    outs = [torch.empty((lines, ncols, 4), dtype=torch.float64, device=cube.device) for _ in range(depth)]
    with FlightlinePipeline(depth, cube.device) as pipe:
        for i in range(depth):
            pipe.submit(cube, lib, out=outs[i], out_column0=0, active=active, reflectance=reflectance)
        pipe.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            pipe.submit(cube, lib, out=outs[i % depth], out_column0=0, active=active, reflectance=reflectance)
        pipe.synchronize()
        dt = (time.perf_counter() - t0) / steps
    del outs
    cmf._Workspace._bufs.clear()
    torch.cuda.empty_cache()
    return {"metric": "CMF Mpixels/s, %s window" % ("reflectance (-R)" if reflectance else "CO2"),
            "value": round(lines * ncols / dt / 1e6, 3), "unit": "Mpixel/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps,
            "in_flight": depth, "dtype": "f64", "per_pixel_vs_ch4": round(dt * 1e3 / ch4_ms, 3),
            "workload": "the benchmark flightline, active window %d..%d (p = %d)%s, 201-point sweep, unimodal"
                        % (a0, a1, p, ", reflectance target abscf - mu" if reflectance else "")}


def wide_section(cube, lib, steps=2, with_cpu=True):
    """The same flightline with the full-band window 1..425 (SURVEY.md 8(d) mode F425; the reference's own -R mode,
    window 5..420, takes the same route): windows wider than 96 bands run the batched-GEMM kernels of cmf_wide.hip.
    One flightline at a time, one untimed + `steps` timed passes.  8(d)'s fp64 work of the eigen-restatement:
    2 L p^2 C (covariance) + 2 L p^2 C (Y = X V) + 2 L p A C (sweep) = 10.7 TFLOP at p = 425, against the fp64 matrix peak."""
    import torch
    from srcfinder_amd import cmf
    lines, bands, ncols = cube.shape
    p, A = bands, 201
    out = torch.empty((lines, ncols, 4), dtype=torch.float64, device=cube.device)
    cmf.robust_mf(cube, lib, out=out, active=(1, bands))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        cmf.robust_mf(cube, lib, out=out, active=(1, bands))
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    flop = 2.0 * lines * p * p * ncols * 2 + 2.0 * lines * p * A * ncols
    parity = None
    if with_cpu:
        # two columns against the faithful oracle at 20000 x 425 (~45 s each on a core, side by side in spawned workers)
        from oracle import pool as OP
        cols = [97, 431]
        res = cmf.robust_mf(cube, lib, out=out, active=(1, bands))
        host = cube.index_select(2, torch.as_tensor(cols, device=cube.device)).cpu().numpy()
        o = OP.oracle_columns(host, np.asarray(lib, np.float64)[:, 2], per_job=1)
        got = out[:, cols, 3].cpu().numpy()
        nod = o["score"] == -9999.0
        ref = o["score"][~nod]
        ok = bool(np.array_equal(got == -9999.0, nod)) and bool(
            np.all(np.abs(got[~nod] - ref) <= 1e-4 * np.abs(ref) + 1e-9 * np.abs(ref).max()))
        ok = ok and bool(np.array_equal(res.alphaidx.cpu().numpy()[cols], o["alphaidx"]))
        parity = {"value": round(lines * len(cols) / o["seconds"] / 1e6, 6), "unit": "Mpixel/s", "cores": int(o["workers"]), "kind": "port",
                  "parity_on_sample": ok, "sample": "columns %s x %d lines x %d bands against the oracle (%d workers, %.0f s): "
                                                    "NODATA placement and alpha index exact, scores 1e-4 relative"
                                                    % (cols, lines, p, o["workers"], o["seconds"])}
        del res, host
    del out
    cmf._Workspace._bufs.clear()
    torch.cuda.empty_cache()
    # the same passes with three flightlines in flight (the headline's depth policy): the eigensolver's ~7000 short launches of
    # one flightline leave room for another's GEMMs
    from srcfinder_amd.inflight import FlightlinePipeline
    depth = 3
    outs = [torch.empty((lines, ncols, 4), dtype=torch.float64, device=cube.device) for _ in range(depth)]
    with FlightlinePipeline(depth, cube.device) as pipe:
        for i in range(depth):
            pipe.submit(cube, lib, out=outs[i], out_column0=0, active=(1, bands))
        pipe.synchronize()
        t0 = time.perf_counter()
        for i in range(depth):
            pipe.submit(cube, lib, out=outs[i], out_column0=0, active=(1, bands))
        pipe.synchronize()
        dt3 = (time.perf_counter() - t0) / depth
    del outs
    cmf._Workspace._bufs.clear()
    torch.cuda.empty_cache()
    sec = {"metric": "CMF Mpixels/s, full-band window", "value": round(lines * ncols / dt / 1e6, 3), "unit": "Mpixel/s",
            "ms_per_step": round(dt * 1e3, 2), "steps": steps, "dtype": "f64",
            "config": {"workload": "the same flightline, active window 1..%d (p = %d), 201-point sweep, unimodal, one "
                                   "flightline in flight" % (bands, p),
                       "host_enqueue_ms_per_step": round(t_enq / steps * 1e3, 2),
                       "three_in_flight": {"ms_per_step": round(dt3 * 1e3, 2), "value": round(lines * ncols / dt3 / 1e6, 3),
                                           "note": "three flightlines in flight, one pass each (the depth of the headline)"}},
            "roofline": {"bound": "mfma", "achieved": round(flop / dt / 1e12, 2), "peak": 78.6, "unit": "TFLOP/s",
                         "frac": round(flop / dt / 78.6e12, 4), "flop_per_step": flop,
                         "note": "SURVEY 8(d): 2Lp^2C + 2Lp^2C + 2LpAC fp64 flops of the eigen-restatement over the whole "
                                 "step (eigensolver, exact-determinant pass and the score kernel included in the time)"}}
    if parity:
        sec["cpu_baseline"] = parity
    return sec


def e2e_section(cube, lib, cmf_seconds, cnn, strip_lines=2500, with_fcn=True):
    """BASELINE config 4 (cube -> CMF -> CNN saliency map) on the benchmark flightline.  `value` is the PARITY path
    (cnn/cnn_pred_pipeline.py:159-181: one 256 x 256 window per pixel), measured: the CMF step as timed above plus the tile
    scorer over a strip of `strip_lines` full-width lines of this flightline's CMF plane -- 2500 lines = 1 495 000 windows,
    exactly one rank's row shard of an 8-GPU run, ~50 s of GPU time -- and only THEN scaled by 8 to the
    11.96 M windows of the flightline (`strip` holds the measured figures as they are).  The reference's own fast mode (FCN shift-and-stitch,
    cnn/fcn_pred_pipeline.py -- "not result-equivalent", cnn/README.md:173-177) is measured over the WHOLE plane and reported
    under `approximate_mode`; it never stands in for the parity figure.  fp32, seeded synthetic weights (no trained
    checkpoint ships with the reference, .MISSING_LARGE_BLOBS)."""
    import torch
    from srcfinder_amd import cmf, cnn as C
    from srcfinder_amd.cnn_weights import synthetic_state_dict
    lines, _, ncols = cube.shape
    net = C.GoogLeNetHIP(synthetic_state_dict(2024), device=cube.device)
    res = cmf.robust_mf(cube, lib)
    plane = res.out[..., 3].to(torch.float32).contiguous()
    # ---- parity path on a strip in the middle of the flightline (the windows reach 128 lines up and down: real context)
    r0 = max(0, min(lines - strip_lines, lines // 2))
    batch = 1024
    ds = C.FlightlineConvolve(plane, "COVID_QC", device=net.device)
    sal = torch.zeros(lines * ncols, dtype=torch.float32, device=net.device)
    t_first, n_strip = r0 * ncols, strip_lines * ncols

    net.calibrate(ds, batch)                                                    # the split route's per-layer activation scales

    def run_strip():
        # one overflow slot per batch, read once at the end; raised batches are scored again on the fp32 matrix cores
        return C.score_tiles(net, ds, t_first, t_first + n_strip, batch, sal, route="split")

    C.score_tiles(net, ds, t_first, t_first + min(strip_lines, 8) * ncols, batch, sal, route="split")   # buffers, code objects, maps (not the strip)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rescued = run_strip()
    torch.cuda.synchronize()
    t_strip = time.perf_counter() - t0
    tiles_per_s = n_strip / t_strip
    t_tiles = lines * ncols / tiles_per_s
    del ds, sal
    # ---- the approximate fast mode over the whole plane
    t_fcn, valid = 0.0, 0.0
    if with_fcn:
        C.fcn_predict_flightline(plane[:512].contiguous(), net=net)              # buffers, code objects
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fsal = C.fcn_predict_flightline(plane, net=net)
        torch.cuda.synchronize()
        t_fcn = time.perf_counter() - t0
        valid = float((fsal != -9999).float().mean().item())
        del fsal
    del plane, res, net
    cmf._Workspace._bufs.clear()
    torch.cuda.empty_cache()
    tot = cmf_seconds + t_tiles
    return {"metric": "CMF + CNN saliency map end to end, one flightline on one GPU (parity path)",
            "value": round(lines * ncols / tot / 1e6, 4), "unit": "Mpixel/s", "seconds": round(tot, 1),
            "cmf_seconds": round(cmf_seconds, 4), "cnn_seconds": round(t_tiles, 1), "dtype": "f32",
            "mode": "per-pixel 256 x 256 tile scorer (cnn_pred_pipeline.py:159-181), batch %d" % batch,
            "measured": "the whole CMF step + %d windows (%d full-width lines from line %d) in %.3f s = %.0f windows/s, "
                        "scaled to the flightline's %d windows" % (n_strip, strip_lines, r0, t_strip, tiles_per_s,
                                                                   lines * ncols),
            "strip": {"lines": strip_lines, "windows": n_strip, "seconds": round(t_strip, 3), "windows_per_s": round(tiles_per_s, 1),
                      "batches_rescored_on_fp32": int(rescued),
                      "note": "measured, not extrapolated: the row shard one of 8 ranks scores (cnn/cnn_pred_pipeline.py:159-189)"},
            "data": "synthetic weights (seeded), the CMF plane of this flightline",
            "approximate_mode": {"mode": "fcn shift-and-stitch (the reference's fast mode; not result-equivalent), fp32, whole plane",
                                 "value": round(lines * ncols / (cmf_seconds + t_fcn) / 1e6, 3), "unit": "Mpixel/s",
                                 "seconds": round(cmf_seconds + t_fcn, 3), "cnn_seconds": round(t_fcn, 3),
                                 "saliency_valid_fraction": round(valid, 4)}}


def cnn_section(res, ntiles, batch, with_cpu):
    """CNN tile scorer (cnn/cnn_pred_pipeline.py:159-189): `ntiles` 256 x 256 windows of the CMF band of the product,
    fp32 (the parity path), seeded synthetic weights of the reference's GoogLeNet (no checkpoint ships with it).
    3.706 GFLOP per window (1.853 GMAC) against the fp32 MFMA peak of the guide (157.3 TFLOP/s).
    CPU baseline: the torch-CPU restatement (oracle/cnn_oracle.py) on all usable cores, a bounded sample of windows."""
    import torch
    from srcfinder_amd import cnn
    from srcfinder_amd.cnn_weights import synthetic_state_dict
    sd = synthetic_state_dict(2024)
    net = cnn.GoogLeNetHIP(sd)
    W = res.out.shape[1]
    rows = max(1, (ntiles + W - 1) // W)
    r0 = min(4000, max(0, res.out.shape[0] - rows))
    plane = res.out[r0:r0 + rows, :, 3].to(torch.float32).contiguous()            # a strip of the flightline
    ds = cnn.FlightlineConvolve(plane, "COVID_QC", device=net.device)
    out = torch.zeros(rows * W, dtype=torch.float32, device=net.device)
    ntiles = rows * W                                # whole image rows: the C-side driver sequences the graph (shared trunk)

    scales = net.calibrate(ds, batch)                # the split route's per-layer activation scales, from this plane

    def run(route="split"):
        return cnn.score_tiles(net, ds, 0, ntiles, batch, out, route=route)

    def timed(route):
        run(route)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = run(route)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, n

    dt_w, _ = timed("winograd")                      # the other float32-tolerance routes, kept on record beside the default
    dt_u, _ = timed("split_unshared")
    dt_c, _ = timed("split_conv3")
    dt, rescued = timed("split")
    tf = ntiles * 3.706e9 / dt / 1e12
    # Flops one window EXECUTES on the shared-trunk split route (srcfinder_amd/csrc/cnn_share.hip, cnn_ring.h; SURVEY Appendix C's MACs x 2):
    # the layers through inception3b shrink to the SIDE columns of the per-window ring (band sharing: the top / bottom rows come from
    # strip maps) -- conv2 at the 252 border positions (0.002 GFLOP), conv3 at 256 of 4096 (0.057), inception3a at 160 of 1024 (0.050 of
    # its 0.318), inception3b at 224 (0.152 of 0.694) as fp16 operand-split products, the strip maps 0.033 per window at 598 columns
    # (conv3 on 8 x 16 x 218 positions, inception3a / 3b on 16 x 8 x 108 per image row), conv1 + maxpool1 at the side of the border on the
    # fp32 vector units (0.007) -- the phase maps cost ~1e-4 of a window per window; inception4a .. 5b + head (1.652 GFLOP) run as they are.
    # Every multiply of the split layers is three fp16 MFMA products.
    split_gflop = 0.002 + 0.057 + 0.050 + 0.152 + 0.033 + 1.652
    exec_fp16_tf = ntiles / dt * 3.0 * split_gflop * 1e9 / 1e12
    sec = {"metric": "CNN tiles/s (GoogLeNet, one 256x256 window per pixel)", "value": round(ntiles / dt, 1), "unit": "tiles/s",
           "dtype": "f32 (split-operand: fp16 hi + lo halves, fp32 accumulate)", "data": "synthetic weights (seeded), CMF plane of this run", "tiles": ntiles, "batch": batch,
           "route": "split, trunk through inception3b shared between the overlapping windows -- phase maps + strip maps of the band rows, per window the rings' side columns; two halves of the rows in flight on two streams (cnn.LANES) (an argument of the call; per-layer activation scales "
                    "calibrated on this plane: 2^%d .. 2^%d; one overflow slot per batch, %d of %d batches re-scored on the fp32 matrix cores)"
                    % (int(np.log2(min(scales))), int(np.log2(max(scales))), rescued, (ntiles + batch - 1) // batch),
           "fp32_mfma_route": {"value": round(ntiles / dt_w, 1), "unit": "tiles/s",
                               "note": "route=\"winograd\": Winograd F(2x2, 3x3) + implicit GEMM on the fp32 matrix cores, every window on "
                                       "its own (the split route's rescue path), same windows"},
           "unshared_split_route": {"value": round(ntiles / dt_u, 1), "unit": "tiles/s",
                                    "note": "route=\"split_unshared\": round 5's form, every window evaluated on its own"},
           "shared_through_conv3_only": {"value": round(ntiles / dt_c, 1), "unit": "tiles/s",
                                         "note": "route=\"split_conv3\": round 6's first form (16 phase maps of the 64 x 64 grid)"},
           "roofline": {"bound": "mfma", "achieved": round(exec_fp16_tf, 1), "peak": 2500.0, "unit": "TFLOP/s",
                        "frac": round(exec_fp16_tf / 2500.0, 4), "executed_gflop_per_tile_fp16_products": round(3.0 * split_gflop, 3),
                        "note": "EXECUTED flops against the pipe they run on: the split layers' multiplies as three "
                                "v_mfma_f32_32x32x16_f16 products each (fp32 operands as fp16 hi + lo halves, fp32 accumulate: the fp32 "
                                "tolerance class, every reference golden at its 1e-4) against the dense fp16 matrix peak; all kernels of "
                                "a forward pass in the time (pools, pool-projections, ring kernels, head included)",
                        "reference_flops": {"flop_per_tile": 3.706e9, "achieved": round(tf, 2), "unit": "TFLOP/s", "fp32_matrix_peak": 157.3,
                                            "frac_of_fp32_matrix_peak": round(tf / 157.3, 4),
                                            "note": "throughput in the REFERENCE's direct-convolution flops (3.706 GFLOP per window) over the "
                                                    "fp32 matrix peak: above 1 because the restatement executes fewer and cheaper flops"}}}
    if with_cpu:
        from oracle import cnn_oracle as O
        cores = usable_cores()
        torch.set_num_threads(cores)
        n = 64
        pl = plane.cpu().numpy()
        O.predict_plane(pl, sd, *cnn.MODEL_NORM["COVID_QC"], batch=16, indices=range(16))     # warm-up
        t0 = time.perf_counter()
        ref = O.predict_plane(pl, sd, *cnn.MODEL_NORM["COVID_QC"], batch=16, indices=range(n))
        t = time.perf_counter() - t0
        got = out[:n].cpu().numpy()
        v = ref != -9999
        ok = bool(np.array_equal(got == -9999, ~v)) and bool(np.allclose(got[v], ref[v], rtol=2e-4, atol=1e-7))
        sec["cpu_baseline"] = {"value": round(n / t, 2), "unit": "tiles/s", "cores": cores, "kind": "port",
                               "sample": "%d windows, torch CPU (%d threads), %.1f s" % (n, cores, t), "parity_on_sample": ok}
    return sec


def kernel_source_sha():
    """sha256 (16 hex) of the score kernel's source: a PMC record is only quoted for the code it was taken from."""
    import hashlib
    h = hashlib.sha256()
    for f in ("cmf_score.hip", "cmf_common.h"):
        h.update(open(os.path.join(ROOT, "srcfinder_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(lines, samples, p, world):
    """HBM bytes per launch of the score kernel from the PMC passes of tools/pmc_traffic.sh (FETCH_SIZE x2 + WRITE_SIZE,
    separate rocprofv3 --pmc runs of this very command).  Counters cannot be read from inside the run, so the figure
    comes from the newest committed record (profiles/rNN_pmc_traffic.json) -- and ONLY when that record was taken from the
    kernel source that is being run (same sha) on the same geometry; otherwise `traffic` stays null."""
    import glob
    if (lines, samples, p, world) != (LINES, SAMPLES, 72, 1):
        return {}
    sha = kernel_source_sha()
    note = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):   # newest round first
        rec = json.load(open(path))
        name = "profiles/" + os.path.basename(path)
        if rec.get("kernel_source_sha") != sha:
            note = {"traffic_note": "%s was taken from a different kernel source: not quoted" % name}
            continue
        k = rec["kernels"].get("k_score")
        if k:
            return {"traffic": k["hbm_bytes_per_launch"],
                    "traffic_source": "%s (rocprofv3 --pmc passes of this command at the same kernel source, sha %s; not "
                                      "measured by this run)" % (name, sha)}
    return note


PROFILER_ENV_MARKERS = ("ROCPROFILER", "ROCPROF", "ROCP_", "ROCTRACER", "HSA_TOOLS_LIB", "ROCTX")


def under_profiler():
    """True when this process runs under rocprofv3 / rocprof (its tool library is preloaded): a child that inherited the
    preload would open a second counter session on the same GPU and write its own CSVs into the same output directory."""
    if any(k.startswith(PROFILER_ENV_MARKERS) for k in os.environ):
        return True
    return "rocprof" in os.environ.get("LD_PRELOAD", "")


def measured_ceiling(score_ms):
    """The best PURE-TRAFFIC form of the score launch on THIS box: tools/microbench/
    score_ceiling (built by __graft_entry__.build()) moves exactly the launch's bytes in the launch's geometry with none of
    its arithmetic -- 72 of 425 bands of every pixel read, the RGB bands read, 32-byte records written -- in eleven forms.
    Run as a child process on its own 20.3 GB cube while this process idles: a DIFFERENT power / clock state from the
    kernel's position behind the sweep, so the figure is an estimate of the traffic bound, good to a few per cent, not a
    hard ceiling (a ratio above 1 says exactly that and is flagged); the in-step figure is `in_step_traffic_ms`."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "microbench", "score_ceiling")
    if not os.path.isfile(exe) or score_ms <= 0:
        return {}
    env = {k: v for k, v in os.environ.items()
           if not k.startswith(PROFILER_ENV_MARKERS) and k not in ("LD_PRELOAD",)}
    try:
        txt = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env).stdout
        last = [l for l in txt.splitlines() if l.startswith("ceiling_ms")][-1].split(None, 3)
        ceil_ms = float(last[1])
    except Exception as e:                                   # a diagnostic: never fail the benchmark line over it
        return {"ceiling_note": "tools/microbench/score_ceiling did not run: %r" % (e,)}
    ratio = ceil_ms / score_ms
    out = {"ceiling_ms": round(ceil_ms, 4), "frac_of_measured_ceiling": round(ratio, 4),
           "ceiling_frac_of_peak": round(3540160000 / (ceil_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "ceiling_source": "tools/microbench/score_ceiling, same box, separate process: best of 11 pure load/store forms of "
                             "the launch's geometry and bytes (%s) -- the best pure-traffic form measured, not a hard bound"
                             % last[3]}
    if ratio > 1.0:
        out["ceiling_note"] = "the kernel beat the stand-alone traffic form: the two were not measured in the same state"
    return out


def in_step_traffic(cube, lib, active, steps=10):
    """the pure-traffic form of the score launch run IN PLACE of k_score inside the step -- same stream,
    same position behind the sweep, same power state (sf_debug_set(1, 200): the launch's loads and record stores with a
    plain sum instead of the filter, no statistics, no metadata) -- timed by the same HIP events as the kernel."""
    import torch
    from srcfinder_amd import _ffi, cmf
    L = _ffi.lib()
    out = torch.empty((cube.shape[0], cube.shape[2], 4), dtype=torch.float64, device=cube.device)
    res = {}
    for name, variant in (("k_score", 0), ("traffic_only", 200)):
        L.sf_debug_set(1, variant)
        try:
            cmf.robust_mf(cube, lib, out=out, active=active)
            torch.cuda.synchronize()
            L.sf_cmf_score_timing(1)
            for _ in range(steps):
                cmf.robust_mf(cube, lib, out=out, active=active)
            torch.cuda.synchronize()
            tot, n = _ffi.C.c_double(0.0), _ffi.C.c_int(0)
            L.sf_cmf_score_timing_read(_ffi.C.byref(tot), _ffi.C.byref(n))
            L.sf_cmf_score_timing(0)
            res[name] = tot.value / max(n.value, 1)
        finally:
            L.sf_debug_set(1, 0)
    return {"in_step_traffic_ms": round(res["traffic_only"], 4), "in_step_kernel_ms": round(res["k_score"], 4),
            "frac_of_in_step_traffic": round(res["traffic_only"] / res["k_score"], 4) if res["k_score"] > 0 else None,
            "in_step_note": "the launch with its arithmetic, filter table, statistics and metadata removed, run in place of "
                            "k_score inside %d steps (one flightline in flight)" % steps}


def hard_spectrum_section(cube, lib, active, depth, steps=10):
    """The same step on a cube whose columns have correlation spectra spanning 3.5 .. 7 decades (the generator of
    tools/validate_wide_spectrum.py on EVERY column): the sweep's cost depends on the rank at which a column's coefficient
    matrix factors (cmf_lowrank.hip), and the benchmark recipe of SURVEY 8(d) puts every column on the cheapest route."""
    import torch
    from srcfinder_amd import cmf
    from srcfinder_amd.inflight import FlightlinePipeline
    lines, bands, ncols = cube.shape
    a0, a1 = active
    p = a1 - a0 + 1
    hard = cube.clone()
    g = torch.Generator(device=cube.device)
    g.manual_seed(4321)
    decades = torch.linspace(3.5, 7.0, ncols).tolist()
    for c in range(ncols):
        if c == ncols // 3:
            continue                                             # the all-NODATA column stays
        q, _ = torch.linalg.qr(torch.randn((p, p), generator=g, device=cube.device, dtype=torch.float64))
        sd = torch.sqrt(torch.exp(torch.linspace(0.0, -decades[c] * 2.302585092994046, p, device=cube.device,
                                                 dtype=torch.float64)))
        x = 10.0 + 0.5 * (torch.randn((lines, p), generator=g, device=cube.device, dtype=torch.float64) * sd) @ q.T
        hard[7:, a0 - 1:a1, c] = x[7:].float()
    routes = cmf.sweep_routes(hard, lib, active=active)
    outs = [torch.empty((lines, ncols, 4), dtype=torch.float64, device=cube.device) for _ in range(depth)]
    with FlightlinePipeline(depth, cube.device) as pipe:
        for i in range(depth):
            pipe.submit(hard, lib, out=outs[i], out_column0=0, active=active)
        pipe.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            pipe.submit(hard, lib, out=outs[i % depth], out_column0=0, active=active)
        pipe.synchronize()
        dt = (time.perf_counter() - t0) / steps
    del outs, hard
    cmf._Workspace._bufs.clear()
    torch.cuda.empty_cache()
    return {"metric": "CMF Mpixels/s, hard-spectrum cube", "value": round(lines * ncols / dt / 1e6, 3), "unit": "Mpixel/s",
            "ms_per_step": round(dt * 1e3, 3), "steps": steps, "in_flight": depth, "sweep_routes": routes,
            "workload": "the benchmark flightline with the active window of every column replaced by a spectrum spanning "
                        "3.5 .. 7 decades (linear over the columns); the headline cube spans ~2.5"}


def ingest_section(cube, lib, res):
    """SURVEY 8(f) N3 / 8(d) "report H2D separately": (i) host -> HBM rate of the staging path (srcfinder_amd/ingest.py:
    active window + RGB bands only, pinned double-buffered, asynchronous) against the same bytes through pageable memory
    and against one pageable copy of ALL bands (what round 3 did); (ii) file -> product seconds through the command line
    (cli_robust_mf) on an ENVI BIL file of this very cube written to tmpfs, the product compared with the resident-cube
    product.  The file is as large as the machine allows (20.3 GB at full size; fewer lines when RAM is short)."""
    import contextlib, io, shutil, tempfile
    import torch
    from srcfinder_amd import cli_robust_mf, cmf, envi, ingest
    lines, bands, ncols = cube.shape
    a0, a1 = cmf.active_window("ch4", False)
    sec = {"metric": "file -> HBM -> product", "unit": "s"}
    # ---- (i) H2D on a bounded sample: 2000 lines of all bands on the host (2.03 GB)
    ns = min(lines, 2000)
    sample = cube[:ns].cpu().numpy()
    torch.cuda.synchronize()
    ingest.stage_cube(sample[:64], (a0, a1))                                  # code paths, pinned allocator warm
    rates = {}
    for name, pinned in (("pinned", True), ("pageable", False)):
        cc = ingest.stage_cube(sample, (a0, a1), pinned=pinned)
        torch.cuda.synchronize()
        rates[name] = cc.stats
        del cc
    t0 = time.perf_counter()
    whole = torch.from_numpy(sample).cuda()
    torch.cuda.synchronize()
    t_whole = time.perf_counter() - t0
    del whole
    sec["h2d"] = {"compact_bytes": rates["pinned"]["bytes"], "bands_moved": rates["pinned"]["bands_moved"],
                  "pinned_GBps": round(rates["pinned"]["GBps"], 2), "pageable_GBps": round(rates["pageable"]["GBps"], 2),
                  "pinned_host_fill_s": round(rates["pinned"]["host_fill_seconds"], 3),
                  "all_bands_pageable_GBps": round(sample.nbytes / t_whole / 1e9, 2),
                  "sample": "%d lines x %d bands x %d samples float32 (%.2f GB on the host); compact = %d bands"
                            % (ns, bands, ncols, sample.nbytes / 1e9, rates["pinned"]["bands_moved"]),
                  "flightline_projection_s": {"compact_pinned": round(lines / ns * rates["pinned"]["seconds"], 2),
                                              "all_bands_pageable": round(lines / ns * t_whole, 2)}}
    del sample
    # ---- (ii) file -> product through the CLI, tmpfs
    tmp_root = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    avail = 0
    for ln in open("/proc/meminfo"):
        if ln.startswith("MemAvailable:"):
            avail = int(ln.split()[1]) * 1024
    free = min(shutil.disk_usage(tmp_root).free, avail)
    line_bytes = bands * ncols * 4 + ncols * 32
    fl = int(min(lines, max(0, (free - (12 << 30)) // line_bytes)))           # keep 12 GB of RAM clear of the files
    if fl < 64:
        sec["file_to_product"] = {"note": "not run: %.1f GB free in %s" % (free / 1e9, tmp_root)}
        return sec
    d = tempfile.mkdtemp(prefix="sf_ingest_", dir=tmp_root)
    try:
        src = os.path.join(d, "flightline_rdn_img")
        envi.write_header(src + ".hdr", {"samples": ncols, "lines": fl, "bands": bands, "header offset": 0,
                                         "file type": "ENVI Standard", "data type": 4, "interleave": "bil", "byte order": 0,
                                         "data ignore value": -9999})
        with open(src, "wb") as f:
            for l0 in range(0, fl, 500):
                cube[l0:min(fl, l0 + 500)].cpu().numpy().tofile(f)
        libpath = os.path.join(d, "lib_ch4_unit.txt")
        np.savetxt(libpath, np.asarray(lib, np.float64))
        dst = os.path.join(d, "flightline_ch4mf_img")
        sink = io.StringIO()
        with contextlib.redirect_stdout(sink):
            cli_robust_mf.main([src, libpath, dst])                           # untimed: code objects, allocator pools
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cli_robust_mf.main(["-v", src, libpath, dst])
            t_cli = time.perf_counter() - t0
        staged = [l for l in sink.getvalue().splitlines() if l.startswith("staged ")]
        prod = np.memmap(dst, dtype=np.float64, mode="r", shape=(fl, ncols, 4))
        if fl == lines:
            want = ingest.fetch_product(res.out)
        else:                                                                  # fewer lines: the statistics differ -> own reference
            want = ingest.fetch_product(cmf.robust_mf(cube[:fl].contiguous(), lib).out)
        same = bool(np.array_equal(prod, want))
        sec["file_to_product"] = {"seconds": round(t_cli, 3), "value": round(fl * ncols / t_cli / 1e6, 3), "unit": "Mpixel/s",
                                  "file_GB": round(fl * bands * ncols * 4 / 1e9, 2), "lines": fl,
                                  "product_bit_identical_to_resident_cube": same, "storage": "tmpfs (%s)" % tmp_root,
                                  "staging": staged[-1] if staged else None,
                                  "command": "python -m srcfinder_amd.cli_robust_mf INPUT lib_ch4_unit.txt OUTPUT "
                                             "(in process, second of two runs)"}
        del prod, want
    finally:
        shutil.rmtree(d, ignore_errors=True)
    cmf._Workspace._bufs.clear()
    torch.cuda.empty_cache()
    return sec


def flush_c_stdio():
    """RCCL prints a version banner to the C stdout of every rank; left in the buffer it would be written at exit, AFTER
    the JSON line.  Flush it early so that the result is the last line of the job's output."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def _cpu_columns(job):
    """Worker of the all-cores baseline (a spawned process: numpy + the oracle only, never torch / HIP)."""
    sub, lib72 = job
    from oracle import cmf_oracle as O
    t0 = time.perf_counter()
    o = O.robust_mf_oracle(sub, lib72, active=(1, sub.shape[1]), rgb_bands=(0, 0, 0))
    return o["out"][..., 3], o["alphaidx"], o["status"], time.perf_counter() - t0


def cpu_baseline(cube, lib, lines, ncols, ncpu_cols, res, active):
    """The oracle (faithful numpy restatement of cmf/robust_mf.py, 201x det+inv+GEMM per column) timed on the host of
    the GPU box over bounded samples of the same cube:
      * ONE core: `ncpu_cols` evenly spaced columns, all lines, in this process;
      * ALL cores: one spawned process per core (OMP_NUM_THREADS=1 each), two columns per core (at most 128 columns);
        wall time from the first submission to the last result (process start-up is outside: the pool is warm).
    Only the active window of the sampled columns travels to the workers."""
    import multiprocessing as mp
    a0, a1 = active
    libw = np.ascontiguousarray(np.asarray(lib, np.float64)[a0 - 1:a1])

    def sample(n):
        cols = [int(round(i * (ncols - 1) / max(n - 1, 1))) for i in range(n)]
        return sorted(set(c for c in cols if c != ncols // 3))       # skip the all-NODATA column (no work)

    def fetch(cols):
        idx = torch_index(cols, cube.device)
        return np.ascontiguousarray(cube[:, a0 - 1:a1, :].index_select(2, idx).cpu().numpy())

    def parity(cols, score, aidx, status):
        got = res.out[:, cols, 3].cpu().numpy()
        nod = score == -9999.0
        ok = bool(np.array_equal(got == -9999.0, nod)) and bool(
            np.all(np.abs(got[~nod] - score[~nod]) <= 1e-4 * np.abs(score[~nod]) + 1e-9 * np.abs(score[~nod]).max()))
        so = status == 0
        return ok and bool(np.array_equal(res.alphaidx.cpu().numpy()[cols][so], aidx[so]))

    # ---- one core
    cols1 = sample(ncpu_cols)
    sc, ai, stt, t1 = _cpu_columns((fetch(cols1), libw))
    one = {"value": round(lines * len(cols1) / t1 / 1e6, 5), "unit": "Mpixel/s", "cores": 1,
           "sample": "%d evenly spaced columns x %d lines of the benchmark cube, %.1f s" % (len(cols1), lines, t1),
           "parity_on_sample": parity(cols1, sc, ai, stt)}
    one["eigen_restatement"] = eigen_restatement_baseline(fetch(cols1), lines, ai, stt)
    # ---- all cores this process may use (affinity mask and cgroup CPU quota: the GPU boxes are containers)
    cores = usable_cores()
    colsn = sample(min(max(2 * cores, 16), 128, ncols - 1))
    sub = fetch(colsn)
    chunks = [list(range(i, len(colsn), cores)) for i in range(min(cores, len(colsn)))]
    ctx = mp.get_context("spawn")                # never fork a process that has initialised HIP
    with ctx.Pool(len(chunks)) as pool:
        pool.map(_warm, range(len(chunks)))      # imports done before the clock starts
        t0 = time.perf_counter()
        outs = pool.map(_cpu_columns, [(np.ascontiguousarray(sub[:, :, ch]), libw) for ch in chunks])
        tn = time.perf_counter() - t0
    score = np.empty((lines, len(colsn)))
    aidx = np.empty(len(colsn), np.int64)
    status = np.empty(len(colsn), np.int32)
    for ch, (s_, a_, st_, _t) in zip(chunks, outs):
        score[:, ch], aidx[ch], status[ch] = s_, a_, st_
    return {"value": round(lines * len(colsn) / tn / 1e6, 5), "unit": "Mpixel/s", "cores": cores, "kind": "port",
            "sample": "%d evenly spaced columns x %d lines of the benchmark cube over %d processes (one per usable host "
                      "core: os.cpu_count() = %d, affinity / cgroup quota = %d; OMP_NUM_THREADS=1), %.1f s wall"
                      % (len(colsn), lines, len(chunks), os.cpu_count() or 1, cores, tn),
            "parity_on_sample": parity(colsn, score, aidx, status),
            "one_core": one}


def eigen_restatement_baseline(sub, lines, aidx_ref, status_ref):
    """SURVEY 8(d)(iii), for information: the alpha selection through ONE symmetric eigendecomposition per column
    (oracle.looshrinkage_eig: the restatement the GPU path is built on) on one host core, the same columns as the faithful
    201 x det + inv + GEMM loop beside it; `alpha_index_equal` compares its argmin with the faithful oracle's."""
    from oracle import cmf_oracle as O
    alphas = O.alpha_grid()
    t0 = time.perf_counter()
    same = True
    for c in range(sub.shape[2]):
        col = sub[:, :, c]
        use = O.useidx(col)
        if status_ref[c] != 0 or use.size < 2:
            continue
        x = np.float64(col[use, :])
        nll = np.empty(len(alphas))
        _cm, mindex = O.looshrinkage_eig(x - x.mean(axis=0), alphas, nll, use.size)
        same = same and int(mindex) == int(aidx_ref[c])
    t = time.perf_counter() - t0
    return {"value": round(lines * sub.shape[2] / t / 1e6, 5), "unit": "Mpixel/s", "cores": 1, "seconds": round(t, 2),
            "alpha_index_equal": bool(same),
            "note": "alpha selection only (mask, mean, one eigh, 201-point sweep as two GEMMs); the filter and the scores are not in it"}


def usable_cores():
    """CPUs this process can actually run on: the affinity mask, capped by the cgroup CPU quota (v2 cpu.max, v1 cfs)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return max(1, n)


def _warm(_):
    from oracle import cmf_oracle  # noqa: F401
    return 0


def torch_index(cols, device):
    import torch
    return torch.as_tensor(cols, dtype=torch.long, device=device)


if __name__ == "__main__":
    main()
