# per-launch times of one tile-scorer batch at several batch sizes (is a layer bound by where its input lives?)
out=gpurun_out/bsweep; mkdir -p $out
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for b in 64 128 256 512 1024; do
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof$b -o p -- python3 $root/tools/bench_cnn.py --tiles $((8*b)) --width 512 --batch $b --lanes 1 --route split > $root/$out/prof$b.log 2>&1
f=$(find $root/$out/prof$b -name "*kernel_trace.csv" | head -1)
python3 $root/tools/cnn_layers.py $f > $root/$out/layers$b.txt 2>&1
tail -1 $root/$out/layers$b.txt
done
