# A/B of one sf_debug_set knob on the per-launch times of a tile-scorer batch: bash tools/experiments/ab_layers.sh KEY=VALUE
out=gpurun_out/ab; mkdir -p $out
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for k in base $1; do
  extra=""; [ "$k" != base ] && extra="--knob $k"
  rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_$k -o p -- python3 $root/tools/bench_cnn.py --tiles 4096 --width 512 --batch 512 --lanes 1 --route split $extra > $root/$out/prof_$k.log 2>&1
  f=$(find $root/$out/prof_$k -name "*kernel_trace.csv" | head -1)
  python3 $root/tools/cnn_layers.py $f > $root/$out/layers_$k.txt 2>&1
done
paste <(awk '{print $1, $2, $4}' $root/$out/layers_base.txt) <(awk '{print $1, $2, $4}' $root/$out/layers_$1.txt)
