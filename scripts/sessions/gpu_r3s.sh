#!/bin/bash
# grid kernel RB = 1 + sparse tiles: register cap (waves per SIMD) sweep
OUT=gpurun_out/r3s
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for w in 4 5 6 8 4 6; do
  PIXIE_MPM_GRID_RB=1 PIXIE_MPM_GRID_WPE=$w PIXIE_MPM_SPARSE=1 PIXIE_MPM_WARM=400 timeout 300 python scripts/mpm_bench.py 1000000 120 2000 2>&1 | grep "^n=" | sed "s/^/rb=1 sparse=1 wpe=$w /" | tee -a $OUT/mpm.txt
done
