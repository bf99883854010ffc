#!/bin/bash
# sparse tiles / RB chosen by the number of active blocks: MPM suite, both bench scenes, the driver's bench command
OUT=gpurun_out/r3u
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
PIXIE_MPM_WARM=400 timeout 300 python scripts/mpm_bench.py 1000000 120 2000 2>&1 | grep "^n=" | tee -a $OUT/mpm.txt
PIXIE_MPM_WARM=400 timeout 300 python scripts/mpm_bench.py 100000 50 4000 2>&1 | grep "^n=" | tee -a $OUT/mpm.txt
timeout 1200 python bench.py --steps 20 --warmup 5 --no-unet-256 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" >> $OUT/bench.err; tail -1 $OUT/bench.err
timeout 900 python -m pytest tests/test_mpm_hip.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_mpm.log 2>&1
grep -E "passed|failed|^E  " $OUT/pytest_mpm.log | tail -5
