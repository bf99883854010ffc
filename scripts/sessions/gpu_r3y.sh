#!/bin/bash
# rocprofv3 --kernel-trace --stats of the MPM step loop alone, per scene (bench.py's profile pools the two scenes)
OUT=gpurun_out/r3y
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$(pwd)
for cfg in "1000000 120 600 1m" "100000 50 2000 100k"; do
  set -- $cfg
  (cd /tmp && PIXIE_MPM_WARM=200 timeout 200 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_$4 -o mpm -- python $ROOT/scripts/mpm_bench.py $1 $2 $3 > $ROOT/$OUT/run_$4.txt 2>&1)
  DB=$(find $OUT/prof_$4 -name "*.db" | head -1)
  [ -n "$DB" ] && python scripts/rocpd_stats.py $DB $OUT/mpm_$4_kernel_stats.csv $OUT/mpm_$4_kernel_stats_by_geometry.csv
  rm -rf $OUT/prof_$4
  grep "^n=" $OUT/run_$4.txt | cut -c1-200; head -4 $OUT/mpm_$4_kernel_stats.csv | cut -c1-170
done
