#!/bin/bash
# wide-window measurement set (GPU box): tests that touch windows wider than 96 bands, kernel stats at p = 425 / 416
tag=${1:-r05wide}; out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests/test_cmf_gpu.py -q -m gpu -x -k "wide or reflectance or full_band or looshrinkage or regulariser or precond or eigensolver or empirical" > $out/pytest_wide.txt 2>&1; tail -4 $out/pytest_wide.txt
B="--no-cpu-baseline --no-cnn --no-e2e --no-wide --no-ingest --no-routes --no-ceiling --no-windows"
tools/prof_bench.sh ${tag}p --active 1,425 --steps 2 --warmup 1 $B --in-flight 1 > $out/prof_wide.log 2>&1
cp gpurun_out/${tag}p_kstats.txt $out/kstats_fullband425.txt; rm -rf gpurun_out/prof_${tag}p
python bench.py --active 1,425 $B --steps 3 --warmup 1 --in-flight 1 > $out/bench_425.json 2> $out/bench.err; cut -c1-330 $out/bench_425.json
python bench.py --active 5,420 $B --steps 3 --warmup 1 --in-flight 1 > $out/bench_416.json 2>> $out/bench.err; cut -c1-330 $out/bench_416.json
sed -n '/timed region/,$p' $out/kstats_fullband425.txt | head -16
