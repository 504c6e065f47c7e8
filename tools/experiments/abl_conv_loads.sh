out=gpurun_out/abl; mkdir -p $out
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for k in 0 64 128 192; do
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof$k -o p -- python3 $root/tools/bench_cnn.py --tiles 2048 --width 512 --batch 512 --lanes 1 --route split --knob 16=$k > $root/$out/prof$k.log 2>&1
f=$(find $root/$out/prof$k -name "*kernel_trace.csv" | head -1)
python3 $root/tools/cnn_layers.py $f > $root/$out/layers$k.txt 2>&1
tail -1 $root/$out/layers$k.txt
done
