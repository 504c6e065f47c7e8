#!/bin/bash
# usage (GPU box, repo root): tools/pmc_bin.sh <tag> "<counters>" <binary> [args]  -> per-kernel averages of the counters
# (one rocprofv3 --pmc pass; the program goes directly after --, no wrapper)
tag=$1; ctrs=$2; shift 2
root=$(pwd)
prog=$(realpath "$1"); shift
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --output-format csv -d $root/gpurun_out/pmc_$tag -o p -- "$prog" "$@" > $root/gpurun_out/pmc_$tag.log 2>&1
cd $root
f=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' | tee gpurun_out/${tag}_pmc.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    acc[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-32s %14.5g  (%d dispatches)" % (c, sum(v.values()) / len(v), len(v)))
PY
rm -rf gpurun_out/pmc_$tag
