#!/bin/bash
# round 4, session a: the product against the reference-code fixture (tests/test_mpm_ref_hip.py), the f3 golden, baseline MPM timings
OUT=gpurun_out/r4a
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_mpm_ref_hip.py tests/test_mpm_hip.py -m gpu -q --tb=short -p no:cacheprovider -s -k "reference_code or export_frame" > $OUT/pytest_ref.log 2>&1
tail -5 $OUT/pytest_ref.log
grep -E "^(jelly|mixed|sand|snow|metal|water|rotation|inverted)" $OUT/pytest_ref.log | cut -c1-600 > $OUT/ref_errors.txt
PIXIE_MPM_WARM=400 timeout 100 python scripts/mpm_bench.py 100000 50 4000 2>&1 | grep "^n=" | tee -a $OUT/mpm.txt
PIXIE_MPM_WARM=200 timeout 200 python scripts/mpm_bench.py 1000000 120 2000 2>&1 | grep "^n=" | tee -a $OUT/mpm.txt
