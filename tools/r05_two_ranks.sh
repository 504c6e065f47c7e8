#!/bin/bash
# The N = 2 path on the one GPU a gpurun box has: two ranks share cuda:0, gloo group, host-staged gather (RCCL refuses two
# ranks on one device).  Functional evidence (the sharded step, the gather, the assembly and its verification run on
# hardware); the value is NOT a scaling number.  -> gpurun_out/r05two/
out=gpurun_out/r05two; mkdir -p $out
python -m pytest tests/test_dist_gpu.py -m gpu -x -q > $out/pytest_dist_gpu.txt 2>&1; tail -3 $out/pytest_dist_gpu.txt
B="--no-cpu-baseline --no-cnn --no-e2e --no-wide --no-ingest --no-routes --no-ceiling --no-windows"
SF_BENCH_SHARE_GPU=1 SF_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 10 --warmup 2 $B > $out/bench_line_two_ranks_one_gpu.json 2> $out/bench_two.err
tail -1 $out/bench_line_two_ranks_one_gpu.json | cut -c1-900; tail -3 $out/bench_two.err
