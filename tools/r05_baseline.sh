#!/bin/bash
# round 5 first measurement: the bench line with the CO2 / -R sections, kernel stats of the CO2 and -R windows (before any change)
out=gpurun_out/r05a; mkdir -p $out
B="--no-cpu-baseline --no-cnn --no-e2e --no-wide --no-ingest --no-routes --no-ceiling --no-windows"
python bench.py --no-cpu-baseline --no-e2e --no-ingest > $out/bench_line.json 2> $out/bench_line.err
tools/prof_bench.sh r05aco2 --active 309,391 --steps 5 --warmup 2 $B --in-flight 1 > $out/prof_co2.log 2>&1
cp gpurun_out/r05aco2_kstats.txt $out/kstats_co2.txt; rm -rf gpurun_out/prof_r05aco2
tools/prof_bench.sh r05arefl --active 5,420 --steps 2 --warmup 1 $B --in-flight 1 > $out/prof_refl.log 2>&1
cp gpurun_out/r05arefl_kstats.txt $out/kstats_refl416.txt; rm -rf gpurun_out/prof_r05arefl
cut -c1-400 $out/bench_line.json
