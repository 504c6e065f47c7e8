#!/bin/bash
# per-rank steps of a 2 / 4 / 8-GPU run on ONE GPU: a rank's own column shard (299 / 150 / 75 of 598), three flightlines in flight and one,
# and the same through a REAL world-1 RCCL gather (SF_BENCH_FORCE_DIST=1: pack, collective, assembly on the slot's stream)
out=gpurun_out/${1:-r05scale}; mkdir -p $out
B="--no-cpu-baseline --no-cnn --no-e2e --no-wide --no-ingest --no-routes --no-ceiling --no-windows --steps 30"
for s in 598 299 150 75; do
  python bench.py --samples $s $B > $out/bench_s$s.json 2>/dev/null
  SF_BENCH_FORCE_DIST=1 python bench.py --samples $s $B > $out/bench_s${s}_dist.json 2>/dev/null
  python3 -c "
import json
a=json.loads(open('$out/bench_s$s.json').read().strip().splitlines()[-1]); b=json.loads(open('$out/bench_s${s}_dist.json').read().strip().splitlines()[-1])
print('samples $s: depth3 %.3f ms, alone %.3f ms | with world-1 RCCL gather: depth3 %.3f ms, alone %.3f ms' % (a['ms_per_step'], a['config']['one_in_flight']['ms_per_step'], b['ms_per_step'], b['config']['one_in_flight']['ms_per_step']))"
done
