#!/bin/bash
# bisect the intermittent wrong results of replayed f16x3 graphs (scripts/unet_soak.py)
OUT=gpurun_out/r3i
mkdir -p $OUT
export TMPDIR=/tmp
run () { echo "== $1"; env $1 timeout 200 python scripts/unet_soak.py 64 128 2 2>&1 | grep -E "MISMATCH|^soak|Error" | tail -6; }
{
run "X=default"
run "PIXIE_UNET_GRAPH=0"
run "PIXIE_UNET_GRAPH_INPLACE=0"
run "PIXIE_UNET_EXECUTOR=python"
run "PIXIE_FUSE_STATS=0"
run "PIXIE_CONV_SPLIT_K=0"
run "PIXIE_FOLD_SKIP=0"
run "PIXIE_FUSE_STATS=0 PIXIE_CONV_SPLIT_K=0 PIXIE_FOLD_SKIP=0"
} > $OUT/bisect.txt 2>&1
cat $OUT/bisect.txt
