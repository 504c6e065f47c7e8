#!/usr/bin/env python3
"""BASELINE configs 4 and 5: cube -> CMF -> CNN saliency map end to end, one flightline per GPU (replicas, no
collective on the data path), per-GPU and aggregate throughput.

    python tools/bench_e2e.py [--lines 20000] [--mode fcn|tiles] [--precision fp32|fp16]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_e2e.py

Every rank generates its own synthetic 598-sample flightline (different seed), runs srcfinder_amd.pipeline.cmf_then_cnn
on it and reports its own seconds; rank 0 prints one JSON line with the per-GPU numbers and the aggregate
(flightline pixels of all ranks / slowest rank's time).  "tiles" is the parity path of the tile scorer (11.96 M windows
per full flightline: minutes -- use --lines to bound it); "fcn" is the reference's own fast mode."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lines", type=int, default=20000)
    ap.add_argument("--samples", type=int, default=598)
    ap.add_argument("--mode", default="fcn", choices=["fcn", "tiles"])
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16"])
    ap.add_argument("--batch", type=int, default=256)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from srcfinder_amd import cnn, pipeline
    from srcfinder_amd.cnn_weights import synthetic_state_dict
    from srcfinder_amd.synth import make_cube_torch

    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    lib = np.load(os.path.join(ROOT, "tests", "golden", "ch4_library.npz"))["library"]
    cube = make_cube_torch(args.lines, args.samples, seed=4321 + rank, abscf_full=lib[:, 2], device=dev)
    net = cnn.GoogLeNetHIP(synthetic_state_dict(2024), device=dev, precision=args.precision)
    small = cube[:256].contiguous()
    pipeline.cmf_then_cnn(small, lib, None, net=net, mode=args.mode, batch=args.batch)     # warm-up, buffers
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    res, sal = pipeline.cmf_then_cnn(cube, lib, None, net=net, mode=args.mode, batch=args.batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ts = torch.tensor([dt], dtype=torch.float64, device=dev)
    allt = [torch.zeros_like(ts) for _ in range(world)]
    if world > 1:
        dist.all_gather(allt, ts)
    else:
        allt = [ts]
    if rank == 0:
        per = [float(t.item()) for t in allt]
        pix = args.lines * args.samples
        print(json.dumps({"metric": "CMF + CNN saliency end to end, one flightline per GPU", "mode": args.mode,
                          "precision": args.precision, "n_gpus": world, "flightline": [args.samples, args.lines, 425],
                          "seconds_per_gpu": [round(x, 3) for x in per],
                          "mpixel_per_s_per_gpu": [round(pix / x / 1e6, 3) for x in per],
                          "aggregate_mpixel_per_s": round(world * pix / max(per) / 1e6, 3),
                          "saliency_valid_fraction": round(float((sal != -9999).float().mean().item()), 4)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
