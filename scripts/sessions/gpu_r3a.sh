#!/bin/bash
# round 3, session a: new MPM paths (tail grid update, packed scatter, wide variant, deferred p2g2p): tests + A/B timings
OUT=gpurun_out/r3a
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mpm_hip.py -m gpu -q -x --tb=short -p no:cacheprovider -s > $OUT/pytest_mpm.log 2>&1
tail -25 $OUT/pytest_mpm.log
for cfg in "100000 50 1000" "1000000 120 300"; do
 for fuse in 0 1; do for bits in 64 32; do
  for wide in 0 1; do
   if [ "$wide" = "1" ] && [ "$cfg" != "100000 50 1000" ]; then continue; fi
   PIXIE_MPM_FUSE=$fuse PIXIE_MPM_BITS=$bits PIXIE_MPM_WIDE=$wide timeout 300 python scripts/mpm_bench.py $cfg 2>&1 | grep "^n=" >> $OUT/variants.log
  done
 done; done
done
PIXIE_MPM_FUSE=0 PIXIE_MPM_PROFILE=2 timeout 300 python scripts/mpm_bench.py 1000000 120 300 2>&1 | grep "^n=" >> $OUT/variants.log
cat $OUT/variants.log
