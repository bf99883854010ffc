#!/bin/bash
# MPM iteration loop on the GPU box: parity tests of the MPM half + timing at both bench sizes.  Usage: gpu_mpm.sh TAG [quick]
TAG=${1:-mpm}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
if [ "$2" != "quick" ]; then
  timeout 900 python -m pytest tests/test_mpm_hip.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest_mpm.log 2>&1
  tail -4 $OUT/pytest_mpm.log
fi
timeout 300 python scripts/mpm_bench.py 1000000 120 300 2>/dev/null | grep "^n=" | tee -a $OUT/mpm_times.txt
timeout 300 python scripts/mpm_bench.py 100000 50 1000 2>/dev/null | grep "^n=" | tee -a $OUT/mpm_times.txt
PIXIE_MPM_TRACE=0x1000 timeout 300 python scripts/mpm_bench.py 1000000 120 300 2>/dev/null | grep "^n=" | tee -a $OUT/mpm_times.txt
