#!/bin/bash
# (Record of an experiment.  The comparison libraries under scripts/_ab/ are not kept: check out the commit named in the matching profiles/ file,
# run `python -m pixie_amd.build`, and copy pixie_amd/libpixie_hip.so there under the name this script expects.)
# Fixed-corotated stress from b - sqrt(b) as a polynomial in E = F F^T - I at small strain (mpm_math.h: fcr_b_minus_sqrt_b), no rotation:
# MPM GPU tests on the new library, same-box alternating timing against the parent commit's library (scripts/_ab/libpixie_hip_prev.so),
# SQ_INSTS_VALU per wave of both.
OUT=gpurun_out/${1:-r4u}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$(pwd)
cp pixie_amd/libpixie_hip.so /tmp/new.so
use () { if [ $1 = new ]; then cp /tmp/new.so pixie_amd/libpixie_hip.so; else cp scripts/_ab/libpixie_hip_$1.so pixie_amd/libpixie_hip.so; fi; }
timeout 900 python -m pytest tests/test_mpm_hip.py tests/test_mpm_ref_hip.py -m gpu -q --tb=short -rA -s -p no:cacheprovider > $OUT/pytest_mpm.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_mpm.log
grep -E "^(PASSED|FAILED|ERROR)|passed|failed" $OUT/pytest_mpm.log | tail -100 > $OUT/pytest_mpm_tail.txt
grep -E "^(sand_rollout|metal_rollout|tree_rollout|jelly_apic|jelly_rpic|jelly_pic|mixed_materials|rotation_release|inverted) (64|32)|hip-vs-f64|config 3" $OUT/pytest_mpm.log | cut -c1-2500 > $OUT/pytest_mpm_numbers.txt
R=$OUT/fcr_series_ab.txt
: > $R
for rep in 1 2 3; do
  for which in new prev; do
    use $which
    echo "== $which (repetition $rep)" >> $R
    PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 120 900 2>&1 | grep "us/substep" | cut -c1-330 >> $R
    PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 100000 50 2200 2>&1 | grep "us/substep" | cut -c1-330 >> $R
    PIXIE_MPM_V0=0.6 PIXIE_MPM_WARM=300 timeout 200 python scripts/mpm_bench.py 1000000 120 600 2>&1 | grep "us/substep" | cut -c1-330 >> $R
  done
done
run_pmc () {  # name, counters..., -- cmd
  local name=$1; shift
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  (cd /tmp && timeout 300 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$name -o $name -- "$@" > $ROOT/$OUT/pmc_$name.log 2>&1)
  local f=$(find $OUT/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python scripts/pmc_summary.py $f > $OUT/pmc_$name.txt; else echo "no counter csv for $name" > $OUT/pmc_$name.txt; tail -5 $OUT/pmc_$name.log >> $OUT/pmc_$name.txt; fi
  rm -rf $OUT/pmc_$name $OUT/pmc_$name.log
}
use new
PIXIE_MPM_WARM=100 run_pmc new_sq SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -- python $ROOT/scripts/mpm_bench.py 1000000 120 60
PIXIE_MPM_V0=0.6 PIXIE_MPM_WARM=300 run_pmc new_sq_in_motion SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -- python $ROOT/scripts/mpm_bench.py 1000000 120 60
grep -E "passed|failed" $OUT/pytest_mpm.log | tail -2; grep FAILED $OUT/pytest_mpm_tail.txt | head; cat $R; grep "mpm_block_kernel<true, true" $OUT/pmc_new_sq*.txt | cut -c1-330
