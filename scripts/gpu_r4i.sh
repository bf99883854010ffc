#!/bin/bash
# round 4, session i: does the step graph help when several scenes share the GPU (host launch rate)?
OUT=gpurun_out/r4i
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for g in 1 0; do
PIXIE_MPM_STEP_GRAPH=$g timeout 300 python scripts/mpm_multi_scene.py 100000 50 2000 1,3,6 2>&1 | grep "^step_graph" | tee -a $OUT/multi.txt
PIXIE_MPM_STEP_GRAPH=$g timeout 300 python scripts/mpm_multi_scene.py 1000000 120 600 2 2>&1 | grep "^step_graph" | tee -a $OUT/multi.txt
done
