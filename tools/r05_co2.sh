#!/bin/bash
# CO2-window measurement set (GPU box): the new route tests, the stage / shard / host-cube tests that touch p = 83, kernel stats, bench line sections
tag=${1:-r05co2}; out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests/test_cmf_gpu.py -q -m gpu -x -k "co2 or unusual_windows or shard or host_cube or stage or golden_S or rank36 or lowrank or sweep_kernels" > $out/pytest_co2.txt 2>&1; tail -5 $out/pytest_co2.txt
B="--no-cpu-baseline --no-cnn --no-e2e --no-wide --no-ingest --no-routes --no-ceiling --no-windows"
tools/prof_bench.sh ${tag}p --active 309,391 --steps 5 --warmup 2 $B --in-flight 1 > $out/prof_co2.log 2>&1
cp gpurun_out/${tag}p_kstats.txt $out/kstats_co2.txt; rm -rf gpurun_out/prof_${tag}p
python bench.py --active 309,391 $B --steps 10 > $out/bench_co2_depth3.json 2> $out/bench_co2.err; cut -c1-330 $out/bench_co2_depth3.json
python bench.py $B --steps 10 > $out/bench_ch4_depth3.json 2>> $out/bench_co2.err; cut -c1-330 $out/bench_ch4_depth3.json
tail -22 $out/kstats_co2.txt
