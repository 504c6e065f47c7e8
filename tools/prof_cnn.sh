#!/bin/bash
# usage (GPU box, repo root): tools/prof_cnn.sh <tag> [bench_cnn.py args...]
tag=$1; shift
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -o p -- python3 $root/tools/bench_cnn.py "$@" > $root/gpurun_out/prof_$tag.log 2>&1
cd $root
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 tools/kstats.py $f > gpurun_out/${tag}_kstats.txt
grep '"metric"' gpurun_out/prof_$tag.log | cut -c1-200
cat gpurun_out/${tag}_kstats.txt
