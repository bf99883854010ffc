#!/bin/bash
# round 4, session b: the reference's driver code against pixie_amd (rows b2 / b5)
OUT=gpurun_out/r4b
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python scripts/reference_drivers/run.py > $OUT/run.log 2>&1
echo "rc=$?" >> $OUT/run.log
cp gpurun_out/reference_drivers.log $OUT/ 2>/dev/null
grep -v "^Particles\|^Total\|^Setting\|^Material ID" $OUT/run.log | tail -40
