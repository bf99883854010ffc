#!/bin/bash
# rocprofv3 kernel trace of the default bench command: per-kernel stats, per-launch-geometry stats, and the JSON line the
# profiled process printed (its conv_kernel_avg_ms is directly comparable with the avg column of kernel_stats.csv).
TAG=${1:-r1p}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --mpm-substeps 300 > $ROOT/$OUT/prof_bench.json 2> $ROOT/$OUT/prof_bench.err)
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python scripts/rocpd_stats.py $DB $OUT/kernel_stats.csv $OUT/kernel_stats_by_geometry.csv
rm -rf $OUT/prof
head -8 $OUT/kernel_stats.csv | cut -c1-150; head -12 $OUT/kernel_stats_by_geometry.csv | cut -c1-170
python -c "import json; b=json.load(open('$OUT/prof_bench.json')); print(json.dumps(b['conv_kernel_avg_ms'], indent=1)); print(b['roofline']['avg_launch_ms'], b['ms_per_step'])"
