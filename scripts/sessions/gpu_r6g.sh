#!/bin/bash
# Round 6, session G: grid-kernel knob sweep at 1 M and on the sand configuration (sparse tiles x loads in flight), diag library (per-kernel event times).
OUT=gpurun_out/${1:-r6g}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$OUT/grid_knobs.txt
: > $R
for sp in 1 0; do for rb in 1 2 4; do
  (PIXIE_MPM_SPARSE=$sp PIXIE_MPM_GRID_RB=$rb PIXIE_MPM_DIAG=1 PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 120 400 2>&1 | grep "us/substep" | sed "s/^/sparse=$sp rb=$rb /" | cut -c1-460) >> $R
  (PIXIE_MPM_SPARSE=$sp PIXIE_MPM_GRID_RB=$rb PIXIE_MPM_DIAG=1 PIXIE_MPM_SCENARIO=sand PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 400 2>&1 | grep "us/substep" | sed "s/^/sparse=$sp rb=$rb /" | cut -c1-460) >> $R
done; done
cut -c1-50,130-150,330-420 $R
