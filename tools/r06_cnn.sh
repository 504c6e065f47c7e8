#!/bin/bash
# Round 6: the tile scorer, shared trunk (route split) against every window on its own (split_unshared), with the per-launch view
out=gpurun_out/${1:-r6cnn}; mkdir -p $out
for r in split split_unshared; do python tools/bench_cnn.py --tiles 16384 --batch 512 --route $r 2>&1 | tail -1 | cut -c1-260; done
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof -o p -- python3 $root/tools/bench_cnn.py --tiles 4784 --batch 512 --route split > $root/$out/prof.log 2>&1
cd $root
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python3 tools/cnn_layers.py $f > $out/cnn_layers.txt 2>&1; cat $out/cnn_layers.txt
