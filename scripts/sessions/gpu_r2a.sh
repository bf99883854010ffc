#!/bin/bash
# Round-2 GPU session A: parity suite (new north-star goldens), MPM occupancy A/B, one bench line.
TAG=${1:-r2a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q --tb=short -rA --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
for occ in 3 4 5; do
  PIXIE_MPM_OCC=$occ timeout 300 python scripts/mpm_bench.py 1000000 120 300 >> $OUT/mpm_ab.txt 2>/dev/null
  PIXIE_MPM_OCC=$occ timeout 300 python scripts/mpm_bench.py 100000 50 1000 >> $OUT/mpm_ab.txt 2>/dev/null
done
PIXIE_MPM_OCC=4 PIXIE_MPM_ITEM_CAP=128 timeout 300 python scripts/mpm_bench.py 1000000 120 300 >> $OUT/mpm_ab.txt 2>/dev/null
PIXIE_MPM_OCC=4 PIXIE_MPM_ITEM_CAP=128 timeout 300 python scripts/mpm_bench.py 100000 50 1000 >> $OUT/mpm_ab.txt 2>/dev/null
timeout 900 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" >> $OUT/bench.err
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3; cat $OUT/mpm_ab.txt; head -c 1500 $OUT/bench.json
