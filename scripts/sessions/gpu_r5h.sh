#!/bin/bash
# Round 5, session H: item_cap 192 / 128 in the DENSE scenes (never measured: round 4 only went upwards from 256), with the XCD-ordered work list.
OUT=gpurun_out/${1:-r5h}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$OUT/item_cap_dense.txt
: > $R
for rep in 1 2; do for cap in 256 192 128; do
  (PIXIE_MPM_ITEM_CAP=$cap PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 120 600 2>&1 | grep "us/substep" | cut -c1-330) >> $R
done; done
for cap in 256 192 128; do
  (PIXIE_MPM_ITEM_CAP=$cap PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 100000 50 3000 2>&1 | grep "us/substep" | cut -c1-330) >> $R
  (PIXIE_MPM_ITEM_CAP=$cap PIXIE_MPM_SCENARIO=metal PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 400 2>&1 | grep "us/substep" | cut -c1-330) >> $R
done
cat $R | cut -c1-300
