#!/bin/bash
# Run a subset of the GPU tests on the GPU box.  Usage: gpu_test.sh TAG "<pytest args>"
TAG=${1:-t}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
eval timeout 1500 python -m pytest "$2" -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1
tail -25 $OUT/pytest.log
