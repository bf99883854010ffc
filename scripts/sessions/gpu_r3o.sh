#!/bin/bash
# deterministic re-binning (slots ranked by (cell, previous slot)) + live_exports: the whole MPM GPU suite, timing check
OUT=gpurun_out/r3o
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_mpm_hip.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_mpm.log 2>&1
grep -E "passed|failed|^E  " $OUT/pytest_mpm.log | tail -8
PIXIE_MPM_WARM=400 timeout 300 python scripts/mpm_bench.py 1000000 120 2000 2>&1 | grep "^n=" | tee -a $OUT/mpm.txt
PIXIE_MPM_WARM=400 timeout 300 python scripts/mpm_bench.py 100000 50 4000 2>&1 | grep "^n=" | tee -a $OUT/mpm.txt
PIXIE_MPM_WARM=400 timeout 300 python scripts/mpm_bench.py 1000000 50 1000 2>&1 | grep "^n=" | tee -a $OUT/mpm.txt
