#!/bin/bash
TAG=${1:-r1n}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python scripts/conv_bench.py 3 > $OUT/conv_bench.log 2>&1
for n in "100000 50 1000" "1000000 120 300"; do
  timeout 200 python scripts/mpm_bench.py $n 2>&1 | grep "^n=" >> $OUT/mpm_bench.log
done
timeout 900 python -m pytest tests/test_unet_hip.py tests/test_mpm_hip.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log; cat $OUT/conv_bench.log | grep cin; cat $OUT/mpm_bench.log
