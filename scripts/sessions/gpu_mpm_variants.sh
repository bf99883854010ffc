#!/bin/bash
TAG=${1:-r1c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for n in "100000 50 600" "1000000 120 200"; do
 for dbg in 0 1 2 3; do
  PIXIE_DEBUG_VARIANT=$dbg timeout 200 python scripts/mpm_bench.py $n 32 2>&1 | grep "^n=" >> $OUT/variants.log
 done
 for cap in 512 1024 2048; do
  PIXIE_ITEM_CAP=$cap timeout 200 python scripts/mpm_bench.py $n 32 2>&1 | grep "^n=" >> $OUT/variants.log
 done
done
timeout 600 python -m pytest tests/test_mpm_hip.py -m gpu -q --tb=short -p no:cacheprovider -k "rollout or boundary" > $OUT/pytest_mpm.log 2>&1
tail -5 $OUT/pytest_mpm.log
cat $OUT/variants.log
