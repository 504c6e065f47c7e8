#!/bin/bash
# development side: after `tools/final_measure.sh <tag>` ran on a GPU box (gpurun merges only gpurun_out/ back), repeat its copies into profiles/
t=${1:-r05}; o=gpurun_out/${t}f
cp $o/pmc_traffic.json profiles/${t}_pmc_traffic.json; cp $o/pmc_mfma.txt profiles/${t}_pmc_mfma_busy.txt; cp $o/pmc_sq.txt profiles/${t}_pmc_sq_waits.txt
cp $o/kstats_inflight1.txt profiles/${t}_bench_kernel_stats_inflight1.txt; cp $o/bench_line_inflight1_rocprof.json profiles/${t}_bench_line_inflight1_rocprof.json
cp $o/bench_line.json profiles/${t}_bench_line.json; cp $o/bench_line_shard75.json profiles/${t}_bench_line_shard75.json
cp $o/bench_line_shard75_inflight1.json profiles/${t}_bench_line_shard75_inflight1.json; cp $o/kstats_fullband425.txt profiles/${t}_fullband425_kernel_stats.txt
cp $o/kstats_co2.txt profiles/${t}_co2_kernel_stats.txt; cp $o/kstats_shard75.txt profiles/${t}_shard75_kernel_stats.txt; cp $o/cnn_layers.txt profiles/${t}_cnn_layers.txt
