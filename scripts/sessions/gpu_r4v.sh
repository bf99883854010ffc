#!/bin/bash
# (Record of an experiment.  The comparison libraries under scripts/_ab/ are not kept: check out the commit named in the matching profiles/ file,
# run `python -m pixie_amd.build`, and copy pixie_amd/libpixie_hip.so there under the name this script expects.)
# Packed fp32 (v_pk_fma_f32) for the x / y components of the regrouped G2P and P2G sums (-DPX_MPM_PK, scripts/_ab/libpixie_hip_pk.so)
# against the shipped scalar kernel: bit-identity of rollouts (sha256 of the state), the MPM GPU tests on the packed build, and a
# same-box alternating timing.
OUT=gpurun_out/${1:-r4v}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cp pixie_amd/libpixie_hip.so /tmp/ship.so
R=$OUT/packed_xy.txt
: > $R
use () { if [ $1 = ship ]; then cp /tmp/ship.so pixie_amd/libpixie_hip.so; else cp scripts/_ab/libpixie_hip_$1.so pixie_amd/libpixie_hip.so; fi; }
for which in ship pk; do
  use $which
  echo "== $which: state hashes" >> $R
  timeout 120 python scripts/mpm_state_hash.py 100000 50 300 2>&1 | grep sha256 >> $R
  timeout 120 python scripts/mpm_state_hash.py 100000 50 300 64 2>&1 | grep sha256 >> $R
  timeout 120 python scripts/mpm_state_hash.py 1000000 120 120 2>&1 | grep sha256 >> $R
done
use pk
timeout 900 python -m pytest tests/test_mpm_hip.py tests/test_mpm_ref_hip.py -m gpu -q --tb=short -x -p no:cacheprovider > $OUT/pytest_mpm_pk.log 2>&1
echo "pytest (packed build) exit $?" >> $R
grep -E "passed|failed" $OUT/pytest_mpm_pk.log | tail -1 >> $R
for rep in 1 2 3; do
  for which in pk ship; do
    use $which
    echo "== $which (repetition $rep)" >> $R
    PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 120 900 2>&1 | grep "us/substep" | cut -c1-330 >> $R
    PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 100000 50 2200 2>&1 | grep "us/substep" | cut -c1-330 >> $R
  done
done
use ship
cat $R
