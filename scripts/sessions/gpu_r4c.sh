#!/bin/bash
# round 4, session c: the whole GPU suite on the tree with the round's parity / ADVICE work
OUT=gpurun_out/r4c
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
tail -15 $OUT/pytest_gpu.log
grep -h "bc test v\|light-side\|frame export vs\|256^3\|middle_block" $OUT/pytest_gpu.log | cut -c1-300
