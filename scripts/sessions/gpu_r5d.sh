#!/bin/bash
# Round 5, session D: material-uniform waves (bin_local_order_kernel orders a block class by class).  Single-material scenes must
# keep their bits (state hash against the round-4 library, jelly: same arithmetic in both); the mixed 1 M scene: timing, SQ_INSTS_VALU
# per wave and kernel durations against session r5a's (82.9 us/substep, 1955 per wave, block kernel 70.6 us); MPM + pipeline GPU tests.
OUT=gpurun_out/${1:-r5d}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$(pwd)
cp pixie_amd/libpixie_hip.so /tmp/new.so
use () { if [ $1 = new ]; then cp /tmp/new.so pixie_amd/libpixie_hip.so; else cp scripts/_ab/libpixie_hip_$1.so pixie_amd/libpixie_hip.so; fi; }
R=$OUT/state_hash.txt
: > $R
for which in new r4; do use $which; echo "== $which" >> $R
  timeout 200 python scripts/mpm_state_hash.py 100000 50 400 2>/dev/null | grep sha256 >> $R
  timeout 200 python scripts/mpm_state_hash.py 1000000 120 200 2>/dev/null | grep sha256 >> $R
  timeout 200 python scripts/mpm_state_hash.py 100000 50 400 64 2>/dev/null | grep sha256 >> $R
done
use new
timeout 900 python -m pytest tests/test_mpm_hip.py tests/test_mpm_ref_hip.py tests/test_pipeline_hip.py -m gpu -q --tb=short -rA -s -p no:cacheprovider > $OUT/pytest_mpm.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_mpm.log
grep -E "^(PASSED|FAILED|ERROR)|passed|failed" $OUT/pytest_mpm.log | tail -120 > $OUT/pytest_mpm_tail.txt
grep -E "@ substep|^mixed|compensated x|mixed_materials" $OUT/pytest_mpm.log | cut -c1-600 > $OUT/pytest_mpm_numbers.txt
T=$OUT/mixed_timing.txt
: > $T
for rep in 1 2; do for sc in mixed metal; do
  (PIXIE_MPM_SCENARIO=$sc PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 400 2>&1 | grep "us/substep" | cut -c1-330) >> $T
done; done
(PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 120 400 2>&1 | grep "us/substep" | cut -c1-330) >> $T
run_pmc () {  # name, counters..., -- cmd
  local name=$1; shift
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  (cd /tmp && timeout 300 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$name -o $name -- "$@" > $ROOT/$OUT/pmc_$name.log 2>&1)
  local f=$(find $OUT/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python scripts/pmc_summary.py $f mpm_ > $OUT/pmc_$name.txt; else echo "no counter csv for $name" > $OUT/pmc_$name.txt; tail -5 $OUT/pmc_$name.log >> $OUT/pmc_$name.txt; fi
  rm -rf $OUT/pmc_$name $OUT/pmc_$name.log
}
run_pmc sq_mixed SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU -- env PIXIE_MPM_SCENARIO=mixed PIXIE_MPM_WARM=100 python $ROOT/scripts/mpm_bench.py 1000000 0 40
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_mixed -o mpm -- env PIXIE_MPM_SCENARIO=mixed PIXIE_MPM_WARM=100 python $ROOT/scripts/mpm_bench.py 1000000 0 300 > $ROOT/$OUT/run_mixed.txt 2>&1)
DB=$(find $OUT/prof_mixed -name "*.db" | head -1)
[ -n "$DB" ] && python scripts/rocpd_stats.py $DB $OUT/stats_mixed.csv
rm -rf $OUT/prof_mixed
cat $R; tail -3 $OUT/pytest_mpm_tail.txt; grep -E "^(FAILED|ERROR)" $OUT/pytest_mpm_tail.txt | head; cat $OUT/pytest_mpm_numbers.txt | cut -c1-300; cat $T | cut -c1-220; grep "true, true" $OUT/pmc_sq_mixed.txt | cut -c1-250; head -3 $OUT/stats_mixed.csv | cut -c1-200
