#!/bin/bash
# usage (GPU box, repo root): tools/pmc_sq_wide.sh <tag>  -> where the waves of the wide-window kernels spend their cycles
# (SQ counters + MFMA busy, counters only, own passes; bench.py --active 1,425)
tag=$1
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
A="--active 1,425 --steps 1 --warmup 1 --no-cpu-baseline --no-cnn --no-e2e --no-wide --no-ingest --no-routes --no-ceiling --in-flight 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $root/gpurun_out/pmc_sqw_$tag -o p -- python3 $root/bench.py $A > $root/gpurun_out/pmc_sqw_$tag.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d $root/gpurun_out/pmc_mfw_$tag -o p -- python3 $root/bench.py $A > $root/gpurun_out/pmc_mfw_$tag.log 2>&1
cd $root
f=$(grep -l "k_score" $(find gpurun_out/pmc_sqw_$tag -name "*counter_collection.csv") | head -1)
g=$(grep -l "k_score" $(find gpurun_out/pmc_mfw_$tag -name "*counter_collection.csv") | head -1)
python3 - "$f" "$g" <<'PY' | tee gpurun_out/${tag}_pmc_wide.txt
import csv, sys, collections, re
def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "anonymous namespace" not in n: continue
        k = n.split("::")[-1].split("(")[0]
        acc[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return acc
sq, mf = load(sys.argv[1]), load(sys.argv[2])
print("# rocprofv3 --pmc (two passes: SQ wave-cycle counters; MFMA busy) -- python3 bench.py --active 1,425 --steps 1 --warmup 1 --in-flight 1 ...")
for k in sorted(sq):
    if not any(x in k for x in ("k_wsweep", "k_wsyrk", "k_blockjac", "k_det_grid", "k_tri", "k_dgemm")): continue
    per = {c: sum(v.values()) / len(v) for c, v in sq[k].items()}
    wc = per.get("SQ_WAVE_CYCLES", 0.0)
    if wc < 1e6: continue
    print(k)
    for c in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_VALU_MFMA_COEXEC_CYCLES"):
        print("   %-28s %12.4g  %5.1f %% of wave cycles" % (c, per.get(c, 0.0), 100.0 * per.get(c, 0.0) / wc))
    d = mf.get(k, {})
    pm = {c: sum(v.values()) / len(v) for c, v in d.items()}
    ga = pm.get("GRBM_GUI_ACTIVE", 0.0)
    if ga:
        print("   matrix pipe busy %.3f   fp64 MFMA wave-instructions per launch %.4g" % (pm.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * ga / 8.0), pm.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)))
PY
rm -rf gpurun_out/pmc_sqw_$tag gpurun_out/pmc_mfw_$tag
