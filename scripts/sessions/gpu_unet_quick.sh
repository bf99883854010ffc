#!/bin/bash
OUT=gpurun_out/${1:-r1x}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_unet_hip.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
for sk in 1 0; do PIXIE_CONV_SPLIT_K=$sk python bench.py --steps 3 --warmup 1 --no-mpm --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('split_k', $sk, round(b['ms_per_step'],2), round(b['unet_conv_ms_per_step'],2), b['layer_ms_top'])"; done
PIXIE_CONV_NSHAPES=12 timeout 300 python scripts/conv_bench.py 3 2>&1 | grep "D=32\|D=16"
