#!/bin/bash
# usage (GPU box, repo root): tools/blockjac_probe.sh SAMPLES  -> median duration of the active k_blockjac launches (full-band window)
root=$(pwd); n=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/bj_$n -o p -- python3 $root/bench.py --no-cpu-baseline --active 1,425 --steps 1 --warmup 0 --samples $n --lines ${2:-20000} > $root/gpurun_out/bj_$n.log 2>&1
cd $root
f=$(find gpurun_out/bj_$n -name "*kernel_trace.csv" | head -1)
python3 - "$f" $n <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'k_blockjac(' in r["Kernel_Name"] or r["Kernel_Name"].endswith('k_blockjac')]
if not rows:
    rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'k_blockjac' in r["Kernel_Name"] and 'flags' not in r["Kernel_Name"] and 'finish' not in r["Kernel_Name"] and 'leftover' not in r["Kernel_Name"]]
d=sorted((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows)
act=[x for x in d if x>15]
print(sys.argv[2], "columns: grid", rows[0]["Grid_Size_X"], rows[0]["Grid_Size_Y"], "active launches", len(act), "median %.0f us"%act[len(act)//2])
PY
