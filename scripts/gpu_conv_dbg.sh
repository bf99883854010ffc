#!/bin/bash
OUT=gpurun_out/${1:-r1t}; mkdir -p $OUT; export TMPDIR=/tmp
for dbg in 0 1 2 3; do echo "dbg=$dbg"; PIXIE_CONV_NO_PIPE=1 PIXIE_CONV_DBG=$dbg PIXIE_CONV_NSHAPES=3 timeout 300 python scripts/conv_bench.py 3 2>&1 | grep cin; done > $OUT/conv_dbg.log
cat $OUT/conv_dbg.log
