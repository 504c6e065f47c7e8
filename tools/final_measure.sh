#!/bin/bash
set -euo pipefail
mkdir -p gpurun_out/r02f
tools/pmc_traffic.sh r02f > gpurun_out/r02f/pmc.log 2>&1
cp gpurun_out/r02f_pmc_traffic.json profiles/r02_pmc_traffic.json
cp gpurun_out/r02f_pmc_traffic.json gpurun_out/r02f/pmc_traffic.json
tools/prof_bench.sh r02fif1 --in-flight 1 --steps 5 --warmup 2 > gpurun_out/r02f/prof_if1.log 2>&1
grep "^{\"metric\"" gpurun_out/prof_r02fif1.log | tail -1 > gpurun_out/r02f/bench_line_inflight1_rocprof.json
cp gpurun_out/r02fif1_kstats.txt gpurun_out/r02f/kstats_inflight1.txt
rm -rf gpurun_out/prof_r02fif1 gpurun_out/pmc_r02f_FETCH_SIZE gpurun_out/pmc_r02f_WRITE_SIZE
python bench.py > gpurun_out/r02f/bench_line.json 2> gpurun_out/r02f/bench_line.err
python bench.py --samples 75 --no-cnn > gpurun_out/r02f/bench_line_shard75.json 2> /dev/null
python bench.py --active 1,425 --steps 3 --warmup 1 --no-cnn --no-cpu-baseline > gpurun_out/r02f/bench_line_fullband425.json 2> /dev/null
tools/prof_bench.sh r02fwide --active 1,425 --steps 2 --warmup 1 --no-cpu-baseline --no-cnn --in-flight 1 > gpurun_out/r02f/prof_wide.log 2>&1
cp gpurun_out/r02fwide_kstats.txt gpurun_out/r02f/kstats_fullband425.txt
rm -rf gpurun_out/prof_r02fwide
root=$(pwd); mkdir -p gpurun_out/r02f/cnnprof; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/r02f/cnnprof -o p -- python3 $root/tools/bench_cnn.py --tiles 1024 --batch 512 > $root/gpurun_out/r02f/cnnprof.log 2>&1
cd $root; t=$(find gpurun_out/r02f/cnnprof -name "*kernel_trace.csv" | head -1); python3 tools/cnn_layers.py $t > gpurun_out/r02f/cnn_layers.txt; rm -rf gpurun_out/r02f/cnnprof
python tools/bench_cnn.py --tiles 8192 --batch 512 > gpurun_out/r02f/cnn_bench_line.json
cut -c1-600 gpurun_out/r02f/bench_line.json; echo; cut -c1-300 gpurun_out/r02f/bench_line_shard75.json; echo; cut -c1-300 gpurun_out/r02f/bench_line_fullband425.json; echo; tail -3 gpurun_out/r02f/cnn_layers.txt
