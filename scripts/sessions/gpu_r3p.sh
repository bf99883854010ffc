#!/bin/bash
# grid kernel: 1 / 2 / 4 node blocks (waves) per workgroup
OUT=gpurun_out/r3p
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_mpm_hip.py -m gpu -q --tb=short -p no:cacheprovider -k "phase or batched or reproducible or bc or grid" > $OUT/pytest_sel.log 2>&1
grep -E "passed|failed|^E  " $OUT/pytest_sel.log | tail -5
for w in 1 2 4 1 4; do
  PIXIE_MPM_GRID_WAVES=$w PIXIE_MPM_WARM=400 timeout 300 python scripts/mpm_bench.py 1000000 120 2000 2>&1 | grep "^n=" | sed "s/^/gw=$w /" | tee -a $OUT/mpm.txt
done
for w in 1 4 2; do
  PIXIE_MPM_GRID_WAVES=$w PIXIE_MPM_WARM=400 timeout 300 python scripts/mpm_bench.py 100000 50 4000 2>&1 | grep "^n=" | sed "s/^/gw=$w /" | tee -a $OUT/mpm.txt
done
