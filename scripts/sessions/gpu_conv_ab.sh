#!/bin/bash
TAG=${1:-convab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PIXIE_CONV_NSHAPES=3
for nb in 0 2 1; do echo "== force_nb $nb"; PIXIE_CONV_FORCE_NB=$nb timeout 300 python scripts/conv_bench.py 5 2>/dev/null | grep cin; done | tee $OUT/conv_ab.txt
