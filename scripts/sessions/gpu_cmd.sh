#!/bin/bash
# Run an arbitrary command line on the GPU box with the usual environment.  Usage: gpu_cmd.sh TAG 'command'
TAG=${1:-cmd}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash -c "$2" 2>&1 | tee $OUT/out.txt | tail -40
