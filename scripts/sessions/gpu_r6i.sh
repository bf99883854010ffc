#!/bin/bash
# Round 6, session I: does the block kernel's scratch frame (404 B/lane, used only by the stale-binning slow path) cost anything when nobody takes the
# path?  A/B against a variant compiled without the slow path (ScratchSize 0, 84 VGPRs): alternating libraries on one box.
OUT=gpurun_out/${1:-r6i}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$OUT/scratch_ab.txt
: > $R
cp pixie_amd/libpixie_hip.so /tmp/new.so
for rep in 1 2 3; do
  for which in shipped noslow; do
    if [ $which = shipped ]; then cp /tmp/new.so pixie_amd/libpixie_hip.so; else cp scripts/_ab/libpixie_hip_noslow.so pixie_amd/libpixie_hip.so; fi
    (PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 120 400 2>&1 | grep "us/substep" | sed "s/^/$which /" | cut -c1-220) >> $R
    (PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 100000 50 3000 2>&1 | grep "us/substep" | sed "s/^/$which /" | cut -c1-220) >> $R
    (PIXIE_MPM_SCENARIO=sand PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 400 2>&1 | grep "us/substep" | sed "s/^/$which /" | cut -c1-220) >> $R
    (PIXIE_MPM_OCC=6 PIXIE_MPM_BITS=64 PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 120 400 2>&1 | grep "us/substep" | sed "s/^/$which occ6-bits64 /" | cut -c1-220) >> $R
  done
done
cp /tmp/new.so pixie_amd/libpixie_hip.so
cut -c1-40,110-135 $R
