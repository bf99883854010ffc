#!/bin/bash
# Round 5, session L: item_cap "auto" (128-thread work items where almost no block holds more than 128 particles).
OUT=gpurun_out/${1:-r5l}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$OUT/item_cap_auto.txt
: > $R
for rep in 1 2; do
  (PIXIE_MPM_SCENARIO=sand PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 400 2>&1 | grep "us/substep" | cut -c1-330) >> $R
  (PIXIE_MPM_ITEM_CAP=256 PIXIE_MPM_SCENARIO=sand PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 400 2>&1 | grep "us/substep" | cut -c1-330) >> $R
done
(PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 120 400 2>&1 | grep "us/substep" | cut -c1-330) >> $R
(PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 200 400 2>&1 | grep "us/substep" | cut -c1-330) >> $R
(PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 100000 50 2000 2>&1 | grep "us/substep" | cut -c1-330) >> $R
(PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 300000 120 600 2>&1 | grep "us/substep" | cut -c1-330) >> $R
for sc in snow metal mixed; do (PIXIE_MPM_SCENARIO=$sc PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 300 2>&1 | grep "us/substep" | cut -c1-330) >> $R; done
cp pixie_amd/libpixie_hip.so /tmp/new.so
for which in new prev; do
  if [ $which = new ]; then cp /tmp/new.so pixie_amd/libpixie_hip.so; else cp scripts/_ab/libpixie_hip_prev.so pixie_amd/libpixie_hip.so; fi
  echo "== $which" >> $OUT/state_hash.txt
  timeout 200 python scripts/mpm_state_hash.py 100000 50 400 2>/dev/null | grep sha256 >> $OUT/state_hash.txt
  timeout 200 python scripts/mpm_state_hash.py 1000000 120 200 2>/dev/null | grep sha256 >> $OUT/state_hash.txt
done
cp /tmp/new.so pixie_amd/libpixie_hip.so
timeout 1200 python -m pytest tests/test_mpm_hip.py tests/test_mpm_ref_hip.py tests/test_pipeline_hip.py tests/test_field_hip.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
cat $R | cut -c1-260; cat $OUT/state_hash.txt; tail -5 $OUT/pytest.log
