#!/bin/bash
# usage (GPU box, repo root): tools/ab_lib.sh KERNEL_SUBSTRING libA.so libB.so ...  -> that kernel's time with each library, same box, interleaved
pat=$1; shift
root=$(pwd)
cp srcfinder_amd/libsrcfinder_amd.so /tmp/lib_orig.so
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for lib in "$@"; do
  cp $root/$lib $root/srcfinder_amd/libsrcfinder_amd.so
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$rep -o p -- python3 $root/bench.py --no-cpu-baseline --steps 10 > /tmp/ab.log 2>&1
  f=$(find /tmp/ab_$rep -name "*kernel_stats.csv" | head -1)
  echo "$lib: $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if '$pat' in r['Name']: print('%.1f us' % (float(r['AverageNs'])/1e3)); break
")  step $(grep metric /tmp/ab.log | grep -o '"ms_per_step": [0-9.]*')"
  rm -rf /tmp/ab_$rep
done
done
cp /tmp/lib_orig.so $root/srcfinder_amd/libsrcfinder_amd.so
