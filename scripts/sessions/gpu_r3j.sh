#!/bin/bash
OUT=gpurun_out/r3j
mkdir -p $OUT
export TMPDIR=/tmp
{ echo "== head zeroed by a kernel"; timeout 300 python scripts/unet_soak.py 64 128 3 2>&1 | grep -E "MISMATCH|^soak|Error" | tail -8; } > $OUT/soak.txt 2>&1
cat $OUT/soak.txt
