#!/bin/bash
# Round 6, session S: knob sweep of the LEAN grid kernel (sparse tiles x tile loads in flight) at 100 k, 300 k, 1 M, sand.
OUT=gpurun_out/${1:-r6s}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$OUT/grid_knobs_lean.txt
: > $R
for sp in 1 0; do for rb in 1 2 4; do
  (PIXIE_MPM_SPARSE=$sp PIXIE_MPM_GRID_RB=$rb PIXIE_MPM_DIAG=1 PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 120 400 2>&1 | grep "us/substep" | sed "s/^/sparse=$sp rb=$rb /" | cut -c1-460) >> $R
  (PIXIE_MPM_SPARSE=$sp PIXIE_MPM_GRID_RB=$rb PIXIE_MPM_DIAG=1 PIXIE_MPM_SCENARIO=sand PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 400 2>&1 | grep "us/substep" | sed "s/^/sparse=$sp rb=$rb /" | cut -c1-460) >> $R
  (PIXIE_MPM_SPARSE=$sp PIXIE_MPM_GRID_RB=$rb PIXIE_MPM_DIAG=1 PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 100000 50 3000 2>&1 | grep "us/substep" | sed "s/^/sparse=$sp rb=$rb /" | cut -c1-460) >> $R
  (PIXIE_MPM_SPARSE=$sp PIXIE_MPM_GRID_RB=$rb PIXIE_MPM_DIAG=1 PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 300000 120 800 2>&1 | grep "us/substep" | sed "s/^/sparse=$sp rb=$rb /" | cut -c1-460) >> $R
done; done
python - <<'PY'
import re
for l in open('gpurun_out/r6s/grid_knobs_lean.txt'):
    m=re.search(r'(sparse=\d rb=\d) n=(\d+) ng=(\d+) .* (tree|sand) .*: ([\d.]+) us/substep.*fused kernel ([\d.]+) us.*grid kernel ([\d.]+) us', l)
    if m: print(m.groups())
PY
