#!/bin/bash
# usage (GPU box, repo root): tools/pmc_cnn.sh <tag> [bench_cnn.py args]  -> where the waves of the tile scorer's kernels spend their cycles
tag=$1; shift
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $root/gpurun_out/pmc_cnn_$tag -o p -- python3 $root/tools/bench_cnn.py --tiles 1024 --batch 512 "$@" > $root/gpurun_out/pmc_cnn_$tag.log 2>&1
cd $root
f=$(find gpurun_out/pmc_cnn_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' | tee gpurun_out/${tag}_pmc_cnn.txt
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
grid = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    m = re.search(r"\d+(k_[a-z0-9_]+)(?:ILi(\d+))?", n) if n.startswith("_Z") else None
    k = (m.group(1) + ("<%s>" % m.group(2) if m.group(2) else "")) if m else n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if not k.startswith("k_"): continue
    key = (k, r.get("Grid_Size", ""))
    acc[key][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
rows = []
for key, d in acc.items():
    per = {c: sum(v.values()) / len(v) for c, v in d.items()}
    wc = per.get("SQ_WAVE_CYCLES", 0.0)
    if wc < 5e7: continue
    rows.append((wc, key, per))
for wc, key, per in sorted(rows, reverse=True)[:12]:
    print("%s grid %s" % key)
    for c in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
        print("   %-24s %12.4g  %5.1f %% of wave cycles" % (c, per.get(c, 0.0), 100.0 * per.get(c, 0.0) / wc))
PY
