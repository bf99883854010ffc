#!/bin/bash
# wide block-kernel variant up to three work items per CU (the 100 k bench scene has 526 items on 256 CUs)
OUT=gpurun_out/r3z
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
PIXIE_MPM_WARM=400 timeout 100 python scripts/mpm_bench.py 100000 50 4000 2>&1 | grep "^n=" | sed "s/^/auto /" | tee -a $OUT/mpm.txt
PIXIE_MPM_WIDE=0 PIXIE_MPM_WARM=400 timeout 100 python scripts/mpm_bench.py 100000 50 4000 2>&1 | grep "^n=" | sed "s/^/wide=0 /" | tee -a $OUT/mpm.txt
timeout 120 python -m pytest tests/test_mpm_hip.py -m gpu -q --tb=short -p no:cacheprovider -k "config3 or reproducible or latency or batched" > $OUT/pytest_sel.log 2>&1
tail -2 $OUT/pytest_sel.log
