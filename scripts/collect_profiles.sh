#!/bin/bash
# Copy the evidence of a gpu_round.sh session from the scratch gpurun_out/TAG into the tracked profiles/ (TAG-prefixed).
TAG=$1
SRC=gpurun_out/$TAG
[ -d "$SRC" ] || { echo "no $SRC"; exit 1; }
for f in kernel_stats.csv kernel_stats_by_geometry.csv mpm_trace_1m.txt mpm_trace_100k.txt kernel_stats_single_stream.csv prof_bench_single_stream.json bench.json prof_bench.json smoke.log device.txt pmc_traffic.json; do
  [ -f $SRC/$f ] && cp $SRC/$f profiles/${TAG}_$f
done
for f in $SRC/pmc_*.txt $SRC/*_microbench.txt $SRC/conv_winograd_bound.txt $SRC/unet_exec_bench.txt; do
  [ -f "$f" ] && cp $f profiles/${TAG}_$(basename $f)
done
[ -f $SRC/pytest_gpu.log ] && tail -45 $SRC/pytest_gpu.log > profiles/${TAG}_pytest_gpu_tail.txt
[ -f $SRC/pmc_traffic.json ] && cp $SRC/pmc_traffic.json profiles/pmc_traffic.json
ls profiles | grep "^${TAG}_" | wc -l
