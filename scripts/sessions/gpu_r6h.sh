#!/bin/bash
# Round 6, session H: the lean grid-kernel instantiation (6 / 8 waves per SIMD, masks one round at a time) on the sand configuration and at 1 M.
OUT=gpurun_out/${1:-r6h}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$OUT/grid_lean.txt
: > $R
for rep in 1 2; do for lean in 0 6 8; do
  (PIXIE_MPM_GRID_LEAN=$lean PIXIE_MPM_DIAG=1 PIXIE_MPM_SCENARIO=sand PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 400 2>&1 | grep "us/substep" | sed "s/^/lean=$lean /" | cut -c1-460) >> $R
done; done
for lean in 0 6 8; do
  (PIXIE_MPM_GRID_LEAN=$lean PIXIE_MPM_DIAG=1 PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 120 400 2>&1 | grep "us/substep" | sed "s/^/lean=$lean /" | cut -c1-460) >> $R
  (PIXIE_MPM_GRID_LEAN=$lean PIXIE_MPM_DIAG=1 PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 200 400 2>&1 | grep "us/substep" | sed "s/^/lean=$lean /" | cut -c1-460) >> $R
done
python - <<'PY'
import re
for l in open('gpurun_out/r6h/grid_lean.txt'):
    m=re.search(r'(lean=\d).* ng=(\d+) .* (tree|sand) .*: ([\d.]+) us/substep.*fused kernel ([\d.]+) us.*grid kernel ([\d.]+) us', l)
    if m: print(m.groups())
PY
