#!/bin/bash
# Round 6, session J: re-binning diet (bin_count with all of a wave's atomics in flight; v / C / F_trial rows not permuted ahead of a G2P), frozen-particle state.
OUT=gpurun_out/${1:-r6j}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$(pwd)
R=$OUT/timing.txt
: > $R
for rep in 1 2; do
(PIXIE_MPM_SCENARIO=sand PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 400 2>&1 | grep "us/substep" | cut -c1-420) >> $R
(PIXIE_MPM_V0=1.0 PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 120 400 2>&1 | grep "us/substep" | cut -c1-420) >> $R
done
(PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 120 400 2>&1 | grep "us/substep" | cut -c1-420) >> $R
(PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 100000 50 3000 2>&1 | grep "us/substep" | cut -c1-420) >> $R
(cd /tmp && PIXIE_MPM_SCENARIO=sand PIXIE_MPM_WARM=100 timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_sand -o mpm -- python $ROOT/scripts/mpm_bench.py 1000000 0 400 > $ROOT/$OUT/run_sand.txt 2>&1)
DB=$(find $OUT/prof_sand -name "*.db" | head -1)
[ -n "$DB" ] && python scripts/rocpd_stats.py $DB $OUT/stats_sand.csv
rm -rf $OUT/prof_sand
timeout 1500 python -m pytest tests/test_mpm_hip.py tests/test_mpm_ref_hip.py tests/test_pipeline_hip.py -m gpu -q --tb=short -p no:cacheprovider -s > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
cut -c1-200 $R; grep -E "bin_|mpm_" $OUT/stats_sand.csv | cut -c1-50,150-; grep "left the grid" $OUT/pytest.log; tail -4 $OUT/pytest.log
