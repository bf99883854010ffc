#!/bin/bash
# round 3, session f: the whole GPU suite + smoke on the current tree
OUT=gpurun_out/r3f
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q --tb=short -rA --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|FAILED|ERROR|^E  " $OUT/pytest_gpu.log | tail -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" >> $OUT/smoke.log; tail -3 $OUT/smoke.log
