#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_traffic.sh <tag>
# HBM traffic per kernel launch of bench.py from the PMC counters, FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (MI355X_MICROARCH.md HBM section), counters only (no trace domains).  Writes gpurun_out/<tag>_pmc_traffic.json.
tag=$1
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $root/gpurun_out/pmc_${tag}_$ctr -o p -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cnn --no-e2e --no-wide --no-ingest --no-routes --no-ceiling --no-windows --in-flight 1 > $root/gpurun_out/pmc_${tag}_$ctr.log 2>&1
done
cd $root
python3 tools/pmc_traffic.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE > gpurun_out/${tag}_pmc_traffic.json
cat gpurun_out/${tag}_pmc_traffic.json
