#!/bin/bash
# usage (GPU box, repo root): tools/gather_trace.sh  -> kernels that are not ours in a forced-dist bench run (75 columns)
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
SF_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/gt -o p -- python3 $root/bench.py --no-cpu-baseline --samples 75 --steps 30 > $root/gpurun_out/gt.log 2>&1
cd $root
f=$(find gpurun_out/gt -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r["Name"]
    if "k_" in n and "anonymous" in n: continue
    print("%-90s calls %6s avg %10.1f us total %8.2f ms" % (n[:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
grep metric gpurun_out/gt.log | grep -o "\"ms_per_step[^,]*"
