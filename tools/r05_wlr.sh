#!/bin/bash
# the wide sweep's rank factorisation (round 5): its test, kernel statistics of the full-band window with and without it
out=gpurun_out/r05wlr; mkdir -p $out
python -m pytest tests/test_cmf_gpu.py -m gpu -x -q -k "wide_sweep_rank_factored" > $out/pytest.txt 2>&1; tail -12 $out/pytest.txt
B="--no-cpu-baseline --no-cnn --no-e2e --no-wide --no-ingest --no-routes --no-ceiling --no-windows"
tools/prof_bench.sh r05wlrF --active 1,425 --steps 2 --warmup 1 $B --in-flight 1 > $out/prof_fact.log 2>&1
cp gpurun_out/r05wlrF_kstats.txt $out/kstats_factored.txt; rm -rf gpurun_out/prof_r05wlrF
tools/prof_bench.sh r05wlrP --active 1,425 --steps 2 --warmup 1 $B --in-flight 1 --knob 24=5 > $out/prof_plain.log 2>&1
cp gpurun_out/r05wlrP_kstats.txt $out/kstats_plain.txt; rm -rf gpurun_out/prof_r05wlrP
head -12 $out/kstats_factored.txt | cut -c1-110; head -6 $out/kstats_plain.txt | cut -c1-110
