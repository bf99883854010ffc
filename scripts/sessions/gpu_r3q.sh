#!/bin/bash
# sparse tile publishing (occupancy masks): bit-identity, the MPM suite with the mode forced on, A/B timing
OUT=gpurun_out/r3q
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_mpm_hip.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_auto.log 2>&1
grep -E "passed|failed|^E  " $OUT/pytest_auto.log | tail -5
PIXIE_MPM_SPARSE_TILES=1 timeout 900 python -m pytest tests/test_mpm_hip.py tests/test_field_transfer_hip.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_sparse1.log 2>&1
grep -E "passed|failed|^E  " $OUT/pytest_sparse1.log | tail -5
for sp in 0 1 0 1; do
  PIXIE_MPM_SPARSE=$sp PIXIE_MPM_WARM=400 timeout 300 python scripts/mpm_bench.py 1000000 120 2000 2>&1 | grep "^n=" | sed "s/^/sparse=$sp /" | tee -a $OUT/mpm.txt
done
for sp in 0 1; do
  PIXIE_MPM_SPARSE=$sp PIXIE_MPM_WARM=400 timeout 300 python scripts/mpm_bench.py 100000 50 4000 2>&1 | grep "^n=" | sed "s/^/sparse=$sp /" | tee -a $OUT/mpm.txt
done
PIXIE_MPM_SPARSE=1 PIXIE_MPM_BITS=64 PIXIE_MPM_WARM=400 timeout 300 python scripts/mpm_bench.py 1000000 120 1000 2>&1 | grep "^n=" | sed "s/^/sparse=1 /" | tee -a $OUT/mpm.txt
