#!/bin/bash
# Round 6, session C: parallel re-binning scan, item_cap decided at the first binning, grid kernel with 4 active blocks per workgroup;
# the hop microbench with the graph variant.
OUT=gpurun_out/${1:-r6c}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$(pwd)
timeout 120 scripts/microbench/stream_hop.exe > $OUT/stream_hop.txt 2>&1
R=$OUT/sand_timing.txt
: > $R
for rep in 1 2; do
  for wpb in 0 1 4; do
    (PIXIE_MPM_GRID_WPB=$wpb PIXIE_MPM_SCENARIO=sand PIXIE_MPM_WARM=100 PIXIE_MPM_DIAG=1 timeout 200 python scripts/mpm_bench.py 1000000 0 400 2>&1 | grep "us/substep" | sed "s/^/wpb=$wpb /" | cut -c1-420) >> $R
  done
done
(PIXIE_MPM_SCENARIO=sand PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 400 2>&1 | grep "us/substep" | cut -c1-420) >> $R
(PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 120 400 2>&1 | grep "us/substep" | cut -c1-420) >> $R
(PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 200 400 2>&1 | grep "us/substep" | cut -c1-420) >> $R
(PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 100000 50 3000 2>&1 | grep "us/substep" | cut -c1-420) >> $R
for sc in snow metal mixed; do (PIXIE_MPM_SCENARIO=$sc PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 300 2>&1 | grep "us/substep" | cut -c1-420) >> $R; done
(cd /tmp && PIXIE_MPM_SCENARIO=sand PIXIE_MPM_WARM=100 timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_sand -o mpm -- python $ROOT/scripts/mpm_bench.py 1000000 0 400 > $ROOT/$OUT/run_sand.txt 2>&1)
DB=$(find $OUT/prof_sand -name "*.db" | head -1)
[ -n "$DB" ] && python scripts/rocpd_stats.py $DB $OUT/stats_sand.csv
rm -rf $OUT/prof_sand
timeout 1500 python -m pytest tests/test_mpm_hip.py tests/test_mpm_ref_hip.py tests/test_pipeline_hip.py tests/test_field_hip.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
cat $OUT/stream_hop.txt | tail -3; cut -c1-20,150-420 $R; grep -E "mpm_|bin_" $OUT/stats_sand.csv | cut -c1-60,150-; tail -5 $OUT/pytest.log
