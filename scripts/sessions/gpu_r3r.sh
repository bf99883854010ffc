#!/bin/bash
# grid kernel: tile loads in flight per candidate block (RB) x sparse tiles
OUT=gpurun_out/r3r
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in "4 0" "1 0" "1 1" "2 1" "2 0" "4 1" "1 1"; do
  set -- $cfg
  PIXIE_MPM_GRID_RB=$1 PIXIE_MPM_SPARSE=$2 PIXIE_MPM_WARM=400 timeout 300 python scripts/mpm_bench.py 1000000 120 2000 2>&1 | grep "^n=" | sed "s/^/rb=$1 sparse=$2 /" | tee -a $OUT/mpm.txt
done
for cfg in "4 0" "2 0" "1 0" "2 1"; do
  set -- $cfg
  PIXIE_MPM_GRID_RB=$1 PIXIE_MPM_SPARSE=$2 PIXIE_MPM_WARM=400 timeout 300 python scripts/mpm_bench.py 100000 50 4000 2>&1 | grep "^n=" | sed "s/^/rb=$1 sparse=$2 /" | tee -a $OUT/mpm.txt
done
