#!/bin/bash
# Upper bound of a 1-D Winograd F(2,3) variant of the f16x3 convolution, measured on the shipped kernel with timing switches
# (pixie_set_option "conv_dbg"): 64 = the tap loop runs 18 of 27 taps (the MFMA count Winograd would have), 2 = only the first
# chunk is staged (no staging cost), 128 = every chunk is staged twice (Winograd's halved tile + input transform roughly double
# the staging work per output).  Results are numerically meaningless under these switches; only the durations matter.
TAG=${1:-wino}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PIXIE_CONV_NSHAPES=3
python -m pixie_amd.build > /dev/null 2>&1
for dbg in 0 64 2 66 128 192; do
  echo "== conv_dbg $dbg"
  PIXIE_CONV_DBG=$dbg timeout 300 python scripts/conv_bench.py 5 2>/dev/null | grep cin | sed -e 's/f32 .*TF)  f16x3/f16x3/' -e 's/speedup.*//'
done | tee $OUT/conv_winograd_bound.txt
