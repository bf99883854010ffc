#!/bin/bash
# grid kernel with the mask words of all rounds fetched first: RB sweep at 1 M, the 100 k scene, bit-identity tests
OUT=gpurun_out/r3w
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for rb in 1 2 4 1 2; do
  PIXIE_MPM_GRID_RB=$rb PIXIE_MPM_WARM=400 timeout 300 python scripts/mpm_bench.py 1000000 120 2000 2>&1 | grep "^n=" | sed "s/^/rb=$rb /" | tee -a $OUT/mpm.txt
done
PIXIE_MPM_WARM=400 timeout 300 python scripts/mpm_bench.py 100000 50 4000 2>&1 | grep "^n=" | sed "s/^/auto /" | tee -a $OUT/mpm.txt
timeout 600 python -m pytest tests/test_mpm_hip.py -m gpu -q --tb=short -p no:cacheprovider -k "sparse or reproducible or 1m or million or full_size or phase" > $OUT/pytest_sel.log 2>&1
grep -E "passed|failed|^E  " $OUT/pytest_sel.log | tail -5
