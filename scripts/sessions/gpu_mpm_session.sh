#!/bin/bash
# MPM-only GPU session: parity tests, timing sweep, rocprofv3 kernel stats -> gpurun_out/$TAG
TAG=${1:-r1b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 900 python -m pytest tests/test_mpm_hip.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_mpm.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_mpm.log
for cfg in "100000 50 1000 32" "100000 50 1000 8" "100000 50 1000 128" "1000000 120 300 32" "1000000 120 300 8"; do
  timeout 300 python scripts/mpm_bench.py $cfg >> $OUT/mpm_bench.log 2>&1
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o mpm -- python $ROOT/scripts/mpm_bench.py 100000 50 500 32 > $ROOT/$OUT/prof_run.log 2>&1)
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python scripts/rocpd_stats.py $DB $OUT/mpm_kernel_stats.csv
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof1m -o mpm -- python $ROOT/scripts/mpm_bench.py 1000000 120 200 32 > $ROOT/$OUT/prof_run1m.log 2>&1)
DB=$(find $OUT/prof1m -name "*.db" | head -1)
[ -n "$DB" ] && python scripts/rocpd_stats.py $DB $OUT/mpm_kernel_stats_1m.csv
rm -rf $OUT/prof $OUT/prof1m
tail -15 $OUT/pytest_mpm.log; cat $OUT/mpm_bench.log; head -12 $OUT/mpm_kernel_stats.csv; head -12 $OUT/mpm_kernel_stats_1m.csv
