#!/bin/bash
# CNN measurement set (GPU box): conv cross-check, parity tests, throughput at batch 512, per-layer kernel trace
tag=${1:-r05cnn}; out=gpurun_out/$tag; mkdir -p $out
python tools/check_conv.py > $out/check_conv.txt 2>&1; tail -3 $out/check_conv.txt
python -m pytest tests/test_cnn_gpu.py -q -m gpu > $out/pytest_cnn.txt 2>&1; tail -3 $out/pytest_cnn.txt
python tools/bench_cnn.py --tiles 4096 --batch 512 > $out/bench_cnn.json 2>&1; tail -1 $out/bench_cnn.json
python tools/bench_cnn.py --tiles 4096 --batch 512 --knob 18=3 > $out/bench_cnn_separate_pool.json 2>&1; echo "18=3 $(tail -1 $out/bench_cnn_separate_pool.json | cut -c60-150)"
root=$(pwd); mkdir -p $out/cnnprof; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $root/$out/cnnprof -o p -- python3 $root/tools/bench_cnn.py --tiles 1024 --batch 512 > $root/$out/cnnprof.log 2>&1
cd $root; t=$(find $out/cnnprof -name "*kernel_trace.csv" | head -1); python3 tools/cnn_layers.py $t > $out/cnn_layers.txt; rm -rf $out/cnnprof
tail -1 $out/cnn_layers.txt
