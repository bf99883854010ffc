#!/bin/bash
# round 3, session e: exact-fp32 variant of the tiled conv body: operator / network parity and timing against the v1 kernel
OUT=gpurun_out/r3e
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_unet_hip.py -m gpu -q --tb=short -p no:cacheprovider -s -k "conv3d_operator or matches_reference_golden or north_star or two_independent or split_k or epilogue_stat" > $OUT/pytest_unet_exact.log 2>&1
grep -v "^$" $OUT/pytest_unet_exact.log | tail -40
echo "== tiled exact kernel"; PIXIE_CONV_NSHAPES=12 timeout 300 python scripts/conv_bench.py 3 2>/dev/null | grep cin | tee $OUT/conv_bench_exact_tiled.txt
echo "== v1 exact kernel"; PIXIE_CONV_EXACT_V1=1 PIXIE_CONV_NSHAPES=12 timeout 300 python scripts/conv_bench.py 3 2>/dev/null | grep cin | tee $OUT/conv_bench_exact_v1.txt
