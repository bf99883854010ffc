#!/bin/bash
# Round-2 GPU session B: parity suite again (field-mapping flow, regenerated config-3 fixture) + block-kernel ablations.
TAG=${1:-r2b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q --tb=short -rA --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
for dbg in 0 0x100 0x200 0x300 0x400 0x800 0xf00; do
  PIXIE_MPM_TRACE=$dbg timeout 300 python scripts/mpm_bench.py 1000000 120 200 2>/dev/null | grep "^n=" >> $OUT/mpm_ablate.txt
done
for dbg in 0 0x100 0x300 0xf00; do
  PIXIE_MPM_TRACE=$dbg timeout 300 python scripts/mpm_bench.py 100000 50 600 2>/dev/null | grep "^n=" >> $OUT/mpm_ablate.txt
done
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -3; grep -E "^FAILED|^E  " $OUT/pytest_gpu.log | head -20; cat $OUT/mpm_ablate.txt
