#!/bin/bash
# round 4, session h: the steady part of the MPM step loop as a captured HIP graph
OUT=gpurun_out/r4h
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_mpm_hip.py tests/test_mpm_ref_hip.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_mpm.log 2>&1
tail -5 $OUT/pytest_mpm.log
for g in 1 0; do
PIXIE_MPM_STEP_GRAPH=$g PIXIE_MPM_WARM=400 timeout 100 python scripts/mpm_bench.py 100000 50 4000 2>&1 | grep "^n=" | sed "s/^/step_graph=$g /" | tee -a $OUT/mpm.txt
PIXIE_MPM_STEP_GRAPH=$g PIXIE_MPM_WARM=200 timeout 200 python scripts/mpm_bench.py 1000000 120 2000 2>&1 | grep "^n=" | sed "s/^/step_graph=$g /" | tee -a $OUT/mpm.txt
done
