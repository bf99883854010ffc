#!/bin/bash
# round 4, session d: regrouped G2P / P2G arithmetic (1491 -> ~1170 VALU instructions per particle-wave): MPM tests + timings
OUT=gpurun_out/r4d
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_mpm_hip.py tests/test_mpm_ref_hip.py tests/test_filling_hip.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_mpm.log 2>&1
tail -6 $OUT/pytest_mpm.log
grep -h "bc test v\|light-side" $OUT/pytest_mpm.log | cut -c1-300
PIXIE_MPM_WARM=400 timeout 100 python scripts/mpm_bench.py 100000 50 4000 2>&1 | grep "^n=" | tee -a $OUT/mpm.txt
PIXIE_MPM_WARM=200 timeout 200 python scripts/mpm_bench.py 1000000 120 2000 2>&1 | grep "^n=" | tee -a $OUT/mpm.txt
PIXIE_MPM_BITS=64 PIXIE_MPM_WARM=200 timeout 200 python scripts/mpm_bench.py 1000000 120 1000 2>&1 | grep "^n=" | sed "s/^/bits64 /" | tee -a $OUT/mpm.txt
