#!/bin/bash
# Round 5, session E: work-item capacity in SPARSE scenes.  The sand configuration (n_grid 200, 1 M particles: 1.9 particles per cell, 64 per
# 4^3 block) runs 11 600 work items of 256 threads with 86 particles on average -- half the waves of a workgroup have no particle.  Does a
# smaller workgroup (set_scalar "item_cap" 64 / 128 / 192) win there, as it did not in dense scenes (HISTORY 3.5: 1 M in 120^3)?
OUT=gpurun_out/${1:-r5e}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$OUT/item_cap_sparse.txt
: > $R
for rep in 1 2; do
  for cap in 256 192 128 64; do
    (PIXIE_MPM_ITEM_CAP=$cap PIXIE_MPM_SCENARIO=sand PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 400 2>&1 | grep "us/substep" | cut -c1-400) >> $R
  done
done
for cap in 256 128; do
  (PIXIE_MPM_ITEM_CAP=$cap PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 200 400 2>&1 | grep "us/substep" | cut -c1-400) >> $R
  (PIXIE_MPM_ITEM_CAP=$cap PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 300000 120 600 2>&1 | grep "us/substep" | cut -c1-400) >> $R
  (PIXIE_MPM_ITEM_CAP=$cap PIXIE_MPM_DIAG=1 PIXIE_MPM_SCENARIO=sand PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 200 2>&1 | grep "us/substep" | cut -c1-400) >> $R
done
cat $R | cut -c1-330
