out=gpurun_out/r6g; mkdir -p $out
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof -o p -- python3 $root/tools/bench_cnn.py --tiles 4096 --width 512 --batch 512 --lanes 1 --route split > $root/$out/prof.log 2>&1
cd $root
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python3 tools/cnn_layers.py $f > $out/cnn_layers.txt 2>&1; cat $out/cnn_layers.txt
