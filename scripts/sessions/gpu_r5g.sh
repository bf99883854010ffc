#!/bin/bash
# Round 5, session G: XCD-aware order of the block kernel's work items (set_scalar "xcd_order"): same bits, timing at 1 M / 100 k / plastic.
OUT=gpurun_out/${1:-r5g}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$OUT/xcd_order.txt
: > $R
for x in 0 1; do PIXIE_MPM_XCD=$x timeout 200 python scripts/mpm_state_hash.py 1000000 120 200 2>/dev/null | grep sha256 >> $R; PIXIE_MPM_XCD=$x timeout 200 python scripts/mpm_state_hash.py 100000 50 400 2>/dev/null | grep sha256 >> $R; done
for rep in 1 2 3; do for x in 0 1; do
  (PIXIE_MPM_XCD=$x PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 120 600 2>&1 | grep "us/substep" | sed "s/^/xcd=$x /" | cut -c1-200) >> $R
done; done
for rep in 1 2; do for x in 0 1; do
  (PIXIE_MPM_XCD=$x PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 100000 50 3000 2>&1 | grep "us/substep" | sed "s/^/xcd=$x /" | cut -c1-200) >> $R
  (PIXIE_MPM_XCD=$x PIXIE_MPM_SCENARIO=metal PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 400 2>&1 | grep "us/substep" | sed "s/^/xcd=$x /" | cut -c1-200) >> $R
done; done
for x in 0 1; do (PIXIE_MPM_XCD=$x PIXIE_MPM_DIAG=1 PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 120 300 2>&1 | grep "us/substep" | sed "s/^/xcd=$x /" | cut -c1-330) >> $R; done
cat $R
