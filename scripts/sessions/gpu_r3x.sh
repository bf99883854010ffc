#!/bin/bash
# last sanity check of the committed binary: build check, smoke, the MPM and C-API GPU tests
OUT=gpurun_out/r3x
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
timeout 600 python -m pytest tests/test_mpm_hip.py tests/test_capi.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1
tail -2 $OUT/pytest.log
