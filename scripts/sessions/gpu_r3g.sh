#!/bin/bash
# round 3, session g: the 256^3 test with per-call diagnostics; momentum conservation with alternating ties; MPM timing
OUT=gpurun_out/r3g
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_unet_hip.py -m gpu -q --tb=short -p no:cacheprovider -s -k "256_cube_128" > $OUT/pytest_unet_256.log 2>&1
grep -E "passed|failed|^E  |finite|f16x3:|f32:|whole networks" $OUT/pytest_unet_256.log | cut -c1-300 | tail -12
timeout 900 python -m pytest tests/test_mpm_hip.py -m gpu -q --tb=short -p no:cacheprovider -s -k "full_size or packed or rollout_parity or plastic_reference" > $OUT/pytest_mpm.log 2>&1
grep -E "passed|failed|^E  |packed scatter" $OUT/pytest_mpm.log | cut -c1-400 | tail -12
for cfg in "1000000 120 2000" "100000 50 2000"; do
  PIXIE_MPM_WARM=400 timeout 300 python scripts/mpm_bench.py $cfg 2>&1 | grep "^n=" >> $OUT/variants.log
done
cat $OUT/variants.log
