#!/bin/bash
TAG=${1:-r1d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 scripts/microbench/lds_atomics.exe > $OUT/lds_atomics.log 2>&1
timeout 900 python -m pytest tests/test_unet_hip.py -m gpu -q --tb=short -p no:cacheprovider -x -s -k "conv3d or golden" > $OUT/pytest_unet.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_unet.log
timeout 600 python scripts/conv_bench.py 3 > $OUT/conv_bench.log 2>&1
cat $OUT/lds_atomics.log; grep -E "rel-L2|passed|failed|Error|error|exit" $OUT/pytest_unet.log | tail -60; cat $OUT/conv_bench.log
