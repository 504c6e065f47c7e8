common="--steps 30 --warmup 3 --no-cpu-baseline --no-cnn --no-wide --no-e2e --no-ceiling --no-ingest --no-routes --no-windows"
mkdir -p gpurun_out/r6s2
for s in 75 598; do
python bench.py $common --samples $s --in-flight 3 > gpurun_out/r6s2/s${s}.log 2>&1
python - gpurun_out/r6s2/s${s}.log $s <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
print("samples %s depth 3: %.3f ms per step (one in flight %.3f)" % (sys.argv[2], d["ms_per_step"], d["config"]["one_in_flight"]["ms_per_step"]))
PY
done
SF_BENCH_FORCE_DIST=1 python bench.py $common --samples 75 --in-flight 3 > gpurun_out/r6s2/s75_rccl1.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r6s2/s75_rccl1.log | head -1 | sed "s/^/samples 75 depth 3 through a world-1 RCCL gather: /"
