#!/bin/bash
# Round 6, session F: A/B of the library with the crushed-frame / rank-deficient paths (inlined) against the commit before them,
# alternating on one box; then the whole GPU suite.
OUT=gpurun_out/${1:-r6f}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$OUT/ab_timing.txt
: > $R
cp pixie_amd/libpixie_hip.so /tmp/new.so
for rep in 1 2; do
  for which in new prev; do
    if [ $which = new ]; then cp /tmp/new.so pixie_amd/libpixie_hip.so; else cp scripts/_ab/libpixie_hip_prev.so pixie_amd/libpixie_hip.so; fi
    (PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 120 400 2>&1 | grep "us/substep" | sed "s/^/$which /" | cut -c1-220) >> $R
    (PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 100000 50 3000 2>&1 | grep "us/substep" | sed "s/^/$which /" | cut -c1-220) >> $R
    for sc in sand metal; do (PIXIE_MPM_SCENARIO=$sc PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 400 2>&1 | grep "us/substep" | sed "s/^/$which /" | cut -c1-220) >> $R; done
  done
done
cp /tmp/new.so pixie_amd/libpixie_hip.so
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
cut -c1-200 $R; tail -6 $OUT/pytest.log
