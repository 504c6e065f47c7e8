#!/bin/bash
out=gpurun_out/${1:-r05cnnx}; mkdir -p $out
python tools/check_conv.py > $out/check_conv.txt 2>&1; tail -1 $out/check_conv.txt
python tools/check_conv.py --knob 17=7 > $out/check_conv7.txt 2>&1; tail -1 $out/check_conv7.txt
for k in 0 7 1 5 6; do python tools/bench_cnn.py --tiles 4096 --batch 512 --knob 17=$k > $out/bench_cnn_knob17_$k.json 2>&1; echo "17=$k $(tail -1 $out/bench_cnn_knob17_$k.json | cut -c60-150)"; done
