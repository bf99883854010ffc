#!/bin/bash
# Round 6, session D: drift target of the re-binning cadence (0.4 / 0.55 / 0.7 cells) on the scenes in motion; full MPM test files
# (parallel scan, first-binning item_cap, product vs float32 oracle on config 3).
OUT=gpurun_out/${1:-r6d}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$OUT/drift_target.txt
: > $R
for d in 0.4 0.55 0.7; do
  (PIXIE_MPM_DRIFT=$d PIXIE_MPM_SCENARIO=sand PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 400 2>&1 | grep "us/substep" | sed "s/^/drift=$d /" | cut -c1-460) >> $R
  (PIXIE_MPM_DRIFT=$d PIXIE_MPM_V0=1.0 PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 120 400 2>&1 | grep "us/substep" | sed "s/^/drift=$d /" | cut -c1-460) >> $R
  (PIXIE_MPM_DRIFT=$d PIXIE_MPM_SCENARIO=ball PIXIE_MPM_V0=3.0 PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 100000 50 2000 2>&1 | grep "us/substep" | sed "s/^/drift=$d /" | cut -c1-460) >> $R
  (PIXIE_MPM_DRIFT=$d PIXIE_MPM_SCENARIO=snow PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 300 2>&1 | grep "us/substep" | sed "s/^/drift=$d /" | cut -c1-460) >> $R
done
timeout 1500 python -m pytest tests/test_mpm_hip.py tests/test_mpm_ref_hip.py tests/test_pipeline_hip.py tests/test_field_hip.py -m gpu -q --tb=short -p no:cacheprovider -s > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
cut -c1-30,100-200,340-460 $R; grep "product vs the FLOAT32" $OUT/pytest.log | cut -c1-400; tail -5 $OUT/pytest.log
