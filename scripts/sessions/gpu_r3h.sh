#!/bin/bash
OUT=gpurun_out/r3h
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python scripts/unet_soak.py 256 128 3 > $OUT/soak_256.txt 2>&1
grep -E "^round|^soak|Error|error" $OUT/soak_256.txt | tail -40
timeout 300 python scripts/unet_soak.py 64 128 4 > $OUT/soak_64.txt 2>&1
grep -E "MISMATCH|^soak|Error|error" $OUT/soak_64.txt | tail -10
