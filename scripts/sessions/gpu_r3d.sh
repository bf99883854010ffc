#!/bin/bash
# round 3, session d: packed scatter scaled by the sum of bounds -- tests + timing; the 256^3 test again
OUT=gpurun_out/r3d
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mpm_hip.py -m gpu -q --tb=short -p no:cacheprovider -s > $OUT/pytest_mpm.log 2>&1
grep -v "^Particles\|^Total\|^Setting\|^Material" $OUT/pytest_mpm.log | grep -v "^\.$" | tail -50
for bits in 64 32; do
  PIXIE_MPM_WARM=400 PIXIE_MPM_BITS=$bits timeout 300 python scripts/mpm_bench.py 1000000 120 2000 2>&1 | grep "^n=" >> $OUT/variants.log
  PIXIE_MPM_WARM=400 PIXIE_MPM_BITS=$bits timeout 300 python scripts/mpm_bench.py 100000 50 2000 2>&1 | grep "^n=" >> $OUT/variants.log
done
cat $OUT/variants.log
timeout 900 python -m pytest tests/test_unet_hip.py -m gpu -q --tb=short -p no:cacheprovider -s -k "256_cube_128" > $OUT/pytest_unet_256.log 2>&1
tail -25 $OUT/pytest_unet_256.log
