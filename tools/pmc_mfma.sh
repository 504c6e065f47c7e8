#!/bin/bash
# usage (GPU box, repo root): tools/pmc_mfma.sh <tag>  -> MFMA-busy evidence per kernel of bench.py (counters only, own pass)
tag=$1
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d $root/gpurun_out/pmc_mfma_$tag -o p -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cnn --no-e2e --no-wide --no-ingest --no-routes --no-ceiling --no-windows --in-flight 1 > $root/gpurun_out/pmc_mfma_$tag.log 2>&1
cd $root
f=$(grep -l "k_score" $(find gpurun_out/pmc_mfma_$tag -name "*counter_collection.csv") | head -1)
python3 - "$f" <<'PY' | tee gpurun_out/${tag}_pmc_mfma.txt
import csv, sys, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "anonymous namespace" not in name: continue
    m = re.search(r"::(k_\w+(?:<[^>]*>)?)", name)
    if not m: continue
    acc[m.group(1)][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cnn --no-e2e --no-wide --no-ingest --no-routes --no-ceiling --no-windows --in-flight 1")
print("# per launch, summed over the device.  busy = MFMA_BUSY_CYCLES / (1024 SIMDs x GUI_ACTIVE / 8 XCDs): the fraction of the kernel's")
print("# cycles in which a SIMD's matrix pipe is executing; MOPS_F64 = fp64 MFMA wave-instructions (512 flop each for v_mfma_f64_4x4x4)")
print("%-22s %8s %16s %14s %16s %8s" % ("kernel", "launches", "MFMA_BUSY_CYCLES", "GUI_ACTIVE", "MFMA_MOPS_F64", "busy"))
rows = []
for k, d in acc.items():
    def per(name):
        v = list(d.get(name, {}).values())
        return (sum(v) / len(v)) if v else 0.0, len(v)
    mb, n = per("SQ_VALU_MFMA_BUSY_CYCLES"); mo, _ = per("SQ_INSTS_VALU_MFMA_MOPS_F64"); ga, _ = per("GRBM_GUI_ACTIVE")
    busy = mb / (1024.0 * ga / 8.0) if ga else 0.0
    rows.append((mb, k, n, ga, mo, busy))
for mb, k, n, ga, mo, busy in sorted(rows, reverse=True):
    if mb == 0 and ga < 1e6: continue
    print("%-22s %8d %16.4g %14.4g %16.4g %8.3f" % (k, n, mb, ga, mo, busy))
PY
