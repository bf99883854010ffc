#!/bin/bash
TAG=${1:-r1f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mpm_hip.py -m gpu -q --tb=short -p no:cacheprovider -s > $OUT/pytest_mpm.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_mpm.log
for n in "100000 50 1000" "1000000 120 300" "100000 64 1000"; do
  timeout 200 python scripts/mpm_bench.py $n 2>&1 | grep "^n=" >> $OUT/mpm_bench.log
done
ROOT=$(pwd)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof1m -o mpm -- python $ROOT/scripts/mpm_bench.py 1000000 120 200 > $ROOT/$OUT/prof_run1m.log 2>&1)
DB=$(find $OUT/prof1m -name "*.db" | head -1)
[ -n "$DB" ] && python scripts/rocpd_stats.py $DB $OUT/mpm_kernel_stats_1m.csv
rm -rf $OUT/prof1m
grep -E "passed|failed|hip-vs|bc test|Error|assert" $OUT/pytest_mpm.log | tail -30; cat $OUT/mpm_bench.log; head -8 $OUT/mpm_kernel_stats_1m.csv
