#!/bin/bash
# follow-up on the final tree: MPM tests with the longer re-binning ceiling, the default bench line (driver's command)
OUT=gpurun_out/r3end2
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_mpm_hip.py tests/test_unet_hip.py -m gpu -q --tb=short -p no:cacheprovider -k "mpm or graph or handle or fused_voxel" > $OUT/pytest_sel.log 2>&1
grep -E "passed|failed|^E  " $OUT/pytest_sel.log | tail -5
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" >> $OUT/bench.err; tail -2 $OUT/bench.err; head -c 1200 $OUT/bench.json; echo
PIXIE_MPM_WARM=400 timeout 300 python scripts/mpm_bench.py 1000000 120 2000 2>&1 | grep "^n=" | tee $OUT/mpm_1m.txt
