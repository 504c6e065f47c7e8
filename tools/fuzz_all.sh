#!/bin/bash
# the fuzz batteries of a round (GPU box): tools/fuzz_all.sh [tag=r04] -> gpurun_out/<tag>_fuzz_summary.txt
root=$(cd "$(dirname "$0")/.." && pwd); cd $root
tag=${1:-r05}; out=gpurun_out/${tag}_fuzz_summary.txt; : > $out
fail=0
run() {   # a failing or timed-out battery fails the script: the exit code of python, not of tail
  echo "== $*" >> $out
  timeout 900 python "$@" > $out.tmp 2>&1; rc=$?
  tail -3 $out.tmp >> $out; rm -f $out.tmp
  if [ $rc -ne 0 ]; then echo "   FAILED (exit $rc)" >> $out; fail=1; fi
}
run tools/fuzz_parity.py 60 701
run tools/fuzz_parity.py 40 702 wide
run tools/fuzz_looshrinkage.py 80 703
run tools/fuzz_wide_full.py 8 704
run tools/fuzz_multimodal.py 40 705
run tools/fuzz_parity.py 50 706 mid
run tools/fuzz_parity.py 12 708 fact
run tools/fuzz_triage.py 707
run tools/fuzz_cnn.py
cat $out
exit $fail
