#!/bin/bash
# the fuzz batteries of a round (GPU box): tools/fuzz_all.sh [tag=r04] -> gpurun_out/<tag>_fuzz_summary.txt
root=$(cd "$(dirname "$0")/.." && pwd); cd $root
tag=${1:-r04}; out=gpurun_out/${tag}_fuzz_summary.txt; : > $out
run() { echo "== $*" >> $out; timeout 900 python "$@" 2>&1 | tail -3 >> $out; }
run tools/fuzz_parity.py 60 701
run tools/fuzz_parity.py 40 702 wide
run tools/fuzz_looshrinkage.py 80 703
run tools/fuzz_wide_full.py 8 704
run tools/fuzz_multimodal.py 40 705
cat $out
