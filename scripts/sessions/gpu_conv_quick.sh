#!/bin/bash
TAG=${1:-r1r}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_unet_hip.py -m gpu -q --tb=short -p no:cacheprovider -x -k "variants or spot_check or conv3d_operator" > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
timeout 600 python scripts/conv_bench.py 3 > $OUT/conv_bench_pipe.log 2>&1
PIXIE_CONV_NO_WS=1 timeout 600 python scripts/conv_bench.py 3 > $OUT/conv_bench_nopipe.log 2>&1
echo "--- ws"; grep cin $OUT/conv_bench_pipe.log; echo "--- plain"; grep cin $OUT/conv_bench_nopipe.log | head -3
