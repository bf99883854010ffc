#!/bin/bash
# (Record of an experiment: the two libraries under scripts/_ab/ were built from the commits named in profiles/r4x_mpm_polar_iteration_variants.txt
# and are not kept.)  Polar iteration (mpm_math.h): Frobenius scaling only far from a rotation, and a one-step exit for nearly rigid particles.  MPM GPU
# tests on the new library, then a same-box A/B/C of the step loop, alternating: new library / scripts/_ab/libpixie_hip_noscale.so (the
# first change only) / scripts/_ab/libpixie_hip_prev.so (built from the parent commit).  PIXIE_MPM_V0=0.6 is a scene in motion (strains of a few per cent).
OUT=gpurun_out/${1:-r4x}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_mpm_hip.py tests/test_mpm_ref_hip.py -m gpu -q --tb=short -rA -s --timeout=900 -p no:cacheprovider > $OUT/pytest_mpm.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_mpm.log
grep -E "^(PASSED|FAILED|ERROR|SKIPPED)|passed|failed" $OUT/pytest_mpm.log | tail -200 > $OUT/pytest_mpm_tail.txt
grep -E "^(tree_rollout|sand_rollout) " $OUT/pytest_mpm.log | cut -c1-1500 > $OUT/pytest_long_rollout_numbers.txt
cp pixie_amd/libpixie_hip.so /tmp/new.so
AB=$OUT/polar_ab.txt
: > $AB
for rep in 1 2; do
  for which in new noscale prev; do
    if [ $which = new ]; then cp /tmp/new.so pixie_amd/libpixie_hip.so; else cp scripts/_ab/libpixie_hip_$which.so pixie_amd/libpixie_hip.so; fi
    echo "== $which (repetition $rep)" >> $AB
    PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 120 900 2>&1 | grep "us/substep" | cut -c1-330 >> $AB
    PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 100000 50 2200 2>&1 | grep "us/substep" | cut -c1-330 >> $AB
    PIXIE_MPM_V0=0.6 PIXIE_MPM_WARM=300 timeout 200 python scripts/mpm_bench.py 1000000 120 600 2>&1 | grep "us/substep" | cut -c1-330 >> $AB
  done
done
cp /tmp/new.so pixie_amd/libpixie_hip.so
grep -E "passed|failed" $OUT/pytest_mpm.log | tail -2; grep FAILED $OUT/pytest_mpm_tail.txt | head; cat $OUT/pytest_long_rollout_numbers.txt | cut -c1-600; cat $AB
