#!/bin/bash
# usage (GPU box, repo root): tools/final_measure.sh <round tag, e.g. r03>
# The round's measurement set: PMC traffic / MFMA-busy / SQ-wait passes, rocprofv3 kernel stats of the one-in-flight bench, the
# default bench line, the 75-column shard line, the wide-window profile, CNN layers.  Everything lands in gpurun_out/<tag>f/;
# copies into profiles/ happen only when every step succeeded (set -e); gpurun merges only gpurun_out/ back, so repeat the
# copies on the development side: for f in ...; see the cp lines at the end.
set -euo pipefail
tag=${1:-r05}
out=gpurun_out/${tag}f
mkdir -p $out
B="--no-cpu-baseline --no-cnn --no-e2e --no-wide --no-ingest --no-routes --no-ceiling --no-windows"
tools/pmc_traffic.sh ${tag}f > $out/pmc.log 2>&1
cp gpurun_out/${tag}f_pmc_traffic.json $out/pmc_traffic.json
cp $out/pmc_traffic.json profiles/${tag}_pmc_traffic.json   # (on the box: the bench runs below quote it when the kernel source sha matches)
tools/pmc_mfma.sh ${tag}f > $out/pmc_mfma.txt 2>&1
tools/pmc_sq.sh ${tag}f k_sweep4s > $out/pmc_sq.txt 2>&1
tools/prof_bench.sh ${tag}fif1 --in-flight 1 --steps 5 --warmup 2 $B > $out/prof_if1.log 2>&1
grep "^{\"metric\"" gpurun_out/prof_${tag}fif1.log | tail -1 > $out/bench_line_inflight1_rocprof.json
cp gpurun_out/${tag}fif1_kstats.txt $out/kstats_inflight1.txt
rm -rf gpurun_out/prof_${tag}fif1 gpurun_out/pmc_${tag}f_FETCH_SIZE gpurun_out/pmc_${tag}f_WRITE_SIZE gpurun_out/pmc_mfma_${tag}f gpurun_out/pmc_sq_${tag}f
python bench.py > $out/bench_line.json 2> $out/bench_line.err
python bench.py --samples 75 --no-cnn > $out/bench_line_shard75.json 2> /dev/null
python bench.py --samples 75 --no-cnn --in-flight 1 --no-cpu-baseline > $out/bench_line_shard75_inflight1.json 2> /dev/null
# a failed step must not reach profiles/: every record has to be what it claims to be (ADVICE r2 / VERDICT r3 item 9)
for j in $out/bench_line.json $out/bench_line_shard75.json $out/bench_line_shard75_inflight1.json $out/bench_line_inflight1_rocprof.json; do
  python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); assert d['value'] > 0 and d['roofline']['avg_launch_ms'] > 0" $j
done
python3 -c "import json,sys; d=json.load(open(sys.argv[1])); assert d['kernels']['k_score']['hbm_bytes_per_launch'] > 1e9" $out/pmc_traffic.json
grep -q "k_score" $out/kstats_inflight1.txt
tools/prof_bench.sh ${tag}fwide --active 1,425 --steps 2 --warmup 1 $B --in-flight 1 > $out/prof_wide.log 2>&1
cp gpurun_out/${tag}fwide_kstats.txt $out/kstats_fullband425.txt
rm -rf gpurun_out/prof_${tag}fwide
tools/prof_bench.sh ${tag}fco2 --active 309,391 --steps 5 --warmup 2 $B --in-flight 1 > $out/prof_co2.log 2>&1
cp gpurun_out/${tag}fco2_kstats.txt $out/kstats_co2.txt
rm -rf gpurun_out/prof_${tag}fco2
tools/prof_bench.sh ${tag}fs75 --samples 75 --steps 10 --warmup 2 $B --in-flight 1 > $out/prof_s75.log 2>&1
cp gpurun_out/${tag}fs75_kstats.txt $out/kstats_shard75.txt
rm -rf gpurun_out/prof_${tag}fs75
root=$(pwd); mkdir -p $out/cnnprof; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $root/$out/cnnprof -o p -- python3 $root/tools/bench_cnn.py --tiles 4096 --width 512 --batch 512 --lanes 1 > $root/$out/cnnprof.log 2>&1
cd $root; t=$(find $out/cnnprof -name "*kernel_trace.csv" | head -1); python3 tools/cnn_layers.py $t > $out/cnn_layers.txt; rm -rf $out/cnnprof
# every step succeeded: the judged copies
cp $out/pmc_traffic.json profiles/${tag}_pmc_traffic.json
cp $out/pmc_mfma.txt profiles/${tag}_pmc_mfma_busy.txt
cp $out/pmc_sq.txt profiles/${tag}_pmc_sq_waits.txt
cp $out/kstats_inflight1.txt profiles/${tag}_bench_kernel_stats_inflight1.txt
cp $out/bench_line_inflight1_rocprof.json profiles/${tag}_bench_line_inflight1_rocprof.json
cp $out/bench_line.json profiles/${tag}_bench_line.json
cp $out/bench_line_shard75.json profiles/${tag}_bench_line_shard75.json
cp $out/bench_line_shard75_inflight1.json profiles/${tag}_bench_line_shard75_inflight1.json
cp $out/kstats_fullband425.txt profiles/${tag}_fullband425_kernel_stats.txt
cp $out/kstats_co2.txt profiles/${tag}_co2_kernel_stats.txt
cp $out/kstats_shard75.txt profiles/${tag}_shard75_kernel_stats.txt
cp $out/cnn_layers.txt profiles/${tag}_cnn_layers.txt
cut -c1-700 $out/bench_line.json; echo; cut -c1-300 $out/bench_line_shard75.json; echo; tail -3 $out/cnn_layers.txt
