#!/bin/bash
# Round 5, session A: the plastic constitutive path after the single-decomposition rewrite (csrc/mpm_math.h: left_stretch +
# return_map_principal) against round 4's library (scripts/_ab/libpixie_hip_r4.so = `python scripts/build_variant.py <r4 commit> ...`).
# MPM GPU tests on the new library; same-box alternating timing of the jelly / sand / snow / metal / mixed 1 M scenes; SQ_INSTS_VALU
# per wave of both; rocprofv3 kernel durations of the new library per scene.
OUT=gpurun_out/${1:-r5a}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$(pwd)
cp pixie_amd/libpixie_hip.so /tmp/new.so
use () { if [ $1 = new ]; then cp /tmp/new.so pixie_amd/libpixie_hip.so; else cp scripts/_ab/libpixie_hip_$1.so pixie_amd/libpixie_hip.so; fi; }
(rocm-smi --showproductname; rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -8) > $OUT/device.txt 2>&1
timeout 900 python -m pytest tests/test_mpm_hip.py tests/test_mpm_ref_hip.py -m gpu -q --tb=short -rA -s -p no:cacheprovider > $OUT/pytest_mpm.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_mpm.log
grep -E "^(PASSED|FAILED|ERROR)|passed|failed" $OUT/pytest_mpm.log | tail -100 > $OUT/pytest_mpm_tail.txt
grep -E "^(sand_rollout|metal_rollout|tree_rollout|jelly_apic|jelly_rpic|jelly_pic|mixed_materials|rotation_release|inverted) (64|32)|hip-vs-f64|config 3|@ substep|inverted particles|^(sand|metal|snow|visplas|water):" $OUT/pytest_mpm.log | cut -c1-2500 > $OUT/pytest_mpm_numbers.txt
R=$OUT/plastic_ab.txt
: > $R
bench () { # scenario
  if [ $1 = jelly ]; then PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 120 400 2>&1 | grep "us/substep" | cut -c1-330
  else PIXIE_MPM_SCENARIO=$1 PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 400 2>&1 | grep "us/substep" | cut -c1-330; fi; }
for rep in 1 2; do
  for which in new r4; do
    use $which
    for sc in jelly sand snow metal mixed; do
      echo "== $which $sc (repetition $rep)" >> $R
      bench $sc >> $R
    done
  done
done
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
scene_cmd () { if [ $1 = jelly ]; then echo "env PIXIE_MPM_WARM=100 python $ROOT/scripts/mpm_bench.py 1000000 120 $2"; else echo "env PIXIE_MPM_SCENARIO=$1 PIXIE_MPM_WARM=100 python $ROOT/scripts/mpm_bench.py 1000000 0 $2"; fi; }
use new
for sc in jelly sand snow metal mixed; do
  run_pmc sq_$sc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU -- $(scene_cmd $sc 40)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_$sc -o mpm -- $(scene_cmd $sc 300) > $ROOT/$OUT/run_$sc.txt 2>&1)
  DB=$(find $OUT/prof_$sc -name "*.db" | head -1)
  [ -n "$DB" ] && python scripts/rocpd_stats.py $DB $OUT/stats_$sc.csv
  rm -rf $OUT/prof_$sc
done
use r4
mkdir -p $OUT/r4
for sc in sand mixed; do
  run_pmc r4_sq_$sc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU -- $(scene_cmd $sc 40)
  mv $OUT/pmc_r4_sq_$sc.txt $OUT/r4/pmc_sq_$sc.txt
done
use new
python scripts/mpm_counters.py $OUT $OUT/mpm_counters.json r5a > /dev/null
tail -3 $OUT/pytest_mpm_tail.txt; cat $R | cut -c1-200; cat $OUT/mpm_counters.json | head -60; cat $OUT/r4/*.txt | cut -c1-300
