#!/bin/bash
# round 4, session f: grid kernel with speculative first-round tile loads; MPM tests + timings
OUT=gpurun_out/r4f
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_mpm_hip.py tests/test_mpm_ref_hip.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_mpm.log 2>&1
tail -4 $OUT/pytest_mpm.log
for i in 1 2; do
PIXIE_MPM_WARM=400 timeout 100 python scripts/mpm_bench.py 100000 50 4000 2>&1 | grep "^n=" | tee -a $OUT/mpm.txt
PIXIE_MPM_WARM=200 timeout 200 python scripts/mpm_bench.py 1000000 120 2000 2>&1 | grep "^n=" | tee -a $OUT/mpm.txt
done
PIXIE_MPM_SPARSE=0 PIXIE_MPM_WARM=200 timeout 200 python scripts/mpm_bench.py 1000000 120 1000 2>&1 | grep "^n=" | sed "s/^/dense-tiles /" | tee -a $OUT/mpm.txt
PIXIE_MPM_SPARSE=1 PIXIE_MPM_WARM=400 timeout 100 python scripts/mpm_bench.py 100000 50 4000 2>&1 | grep "^n=" | sed "s/^/sparse-tiles /" | tee -a $OUT/mpm.txt
