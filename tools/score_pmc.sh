#!/bin/bash
# usage (GPU box, repo root): tools/score_pmc.sh VARIANT [LPW]  -> FETCH_SIZE / WRITE_SIZE of the score kernel alone
root=$(pwd); v=$1; l=${2:-0}
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $root/gpurun_out/spmc_${v}_$ctr -o p -- python3 $root/tools/score_probe.py $v $l > $root/gpurun_out/spmc_${v}_$ctr.log 2>&1
  f=$(find $root/gpurun_out/spmc_${v}_$ctr -name "*counter_collection.csv" | head -1)
  python3 - "$f" $ctr <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("void (anonymous namespace)::k_score") or "k_score" in r["Kernel_Name"]]
vals = {}
for r in rows:
    vals.setdefault(r["Dispatch_Id"], 0.0)
    vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
v = list(vals.values())
print(sys.argv[2], "per launch (KB): mean %.0f over %d launches" % (sum(v) / len(v), len(v)))
PY
done
grep variant $root/gpurun_out/spmc_${v}_FETCH_SIZE.log
