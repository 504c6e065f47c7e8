#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_bench.sh <tag> [bench.py args...]
# rocprofv3 kernel trace + stats of bench.py; condensed table -> gpurun_out/<tag>_kstats.txt
tag=$1; shift
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -o p -- python3 $root/bench.py "$@" > $root/gpurun_out/prof_$tag.log 2>&1
cd $root
# (the trace of the bench.py process: the one with k_score launches; bench.py starts no child under a profiler any more)
t=$(grep -l "k_score" $(find gpurun_out/prof_$tag -name "*kernel_trace.csv") | head -1)
f=${t%kernel_trace.csv}kernel_stats.csv
steps=5; prev=""; for a in "$@"; do if [ "$prev" = "--steps" ]; then steps=$a; fi; prev=$a; done
python3 tools/kstats.py $f $t $steps > gpurun_out/${tag}_kstats.txt
tail -1 gpurun_out/prof_$tag.log | cut -c1-160
cat gpurun_out/${tag}_kstats.txt
