#!/bin/bash
# usage (GPU box, repo root): tools/pmc_sq.sh <tag> [kernel substring]  -> where the waves of a kernel spend their cycles (SQ counters, own pass)
tag=$1; pat=${2:-k_sweep4r}
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $root/gpurun_out/pmc_sq_$tag -o p -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cnn --no-e2e --no-wide --no-ingest --no-routes --no-ceiling --no-windows --in-flight 1 > $root/gpurun_out/pmc_sq_$tag.log 2>&1
cd $root
f=$(grep -l "k_score" $(find gpurun_out/pmc_sq_$tag -name "*counter_collection.csv") | head -1)
python3 - "$f" "$pat" <<'PY' | tee gpurun_out/${tag}_pmc_sq.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "anonymous namespace" not in n: continue
    k = n.split("::")[-1].split("(")[0]
    acc[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for k, d in acc.items():
    if sys.argv[2] not in k and "k_syrk4<" not in k and "k_eigh" not in k and "k_extract_pipe" not in k and "k_score<" not in k: continue
    per = {c: sum(v.values()) / len(v) for c, v in d.items()}
    wc = per.get("SQ_WAVE_CYCLES", 0.0)
    if wc < 1e6: continue
    print(k)
    for c in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_VALU_MFMA_COEXEC_CYCLES"):
        print("   %-28s %12.4g  %5.1f %% of wave cycles" % (c, per.get(c, 0.0), 100.0 * per.get(c, 0.0) / wc))
PY
