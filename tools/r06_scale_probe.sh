#!/bin/bash
# Round 6: the shard step at every shard width of N = 8 / 4 / 2 / 1 with 3, 4 and 6 flightlines in flight,
# and through a real world-1 RCCL gather.  Writes one JSON line per run under gpurun_out/<tag>/.
out=gpurun_out/${1:-r6scale}; mkdir -p $out
common="--steps 30 --warmup 3 --no-cpu-baseline --no-cnn --no-wide --no-e2e --no-ceiling --no-ingest --no-routes --no-windows"
for s in 75 150 299 598; do
  for d in 3 4 6; do
    python bench.py $common --samples $s --in-flight $d > $out/s${s}_d${d}.log 2>&1
    python - "$out/s${s}_d${d}.log" $s $d <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith('{"metric"')]
if not line:
    print("samples %s depth %s: FAILED" % (sys.argv[2], sys.argv[3])); sys.exit(0)
d = json.loads(line[-1])
print("samples %4s depth %s: %.3f ms per step (one in flight %.3f), score kernel %.4f ms" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["config"]["one_in_flight"]["ms_per_step"], d["roofline"]["avg_launch_ms"]))
PY
  done
done
for d in 3 4; do
  SF_BENCH_FORCE_DIST=1 python bench.py $common --samples 75 --in-flight $d > $out/s75_d${d}_rccl1.log 2>&1
  grep -o '"ms_per_step": [0-9.]*' $out/s75_d${d}_rccl1.log | head -1 | sed "s/^/samples 75 depth $d through a world-1 RCCL gather: /"
done
