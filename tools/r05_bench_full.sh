#!/bin/bash
out=gpurun_out/${1:-r05bench}; mkdir -p $out
( time python bench.py > $out/bench_line.json 2> $out/bench_line.err ) 2> $out/bench_time.txt; tail -3 $out/bench_time.txt; cut -c1-300 $out/bench_line.json
for b in 256 512 1024 2048; do python tools/bench_cnn.py --tiles 8192 --batch $b > $out/bench_cnn_b$b.json 2>&1; echo "batch $b $(tail -1 $out/bench_cnn_b$b.json | cut -c60-140)"; done
