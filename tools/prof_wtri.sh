#!/bin/bash
# per-kernel times of the tridiagonal preconditioner on a group of matrices: tools/prof_wtri.sh [p=425] [nb=150]
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/prof_wtri
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_wtri -o p -- python3 $root/tools/check_wtri.py ${1:-425} ${2:-150} 2000 2>&1 | grep "ms per call" | cut -c1-80
python3 - <<PY
import csv, glob
f = glob.glob("$root/gpurun_out/prof_wtri/**/p_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-70s calls %4s  avg %10.1f us  total %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
