#!/bin/bash
# round 3, session b: MPM scatter modes in steady state (long timed regions, rocprofv3 kernel trace), the new U-Net tests,
# the default bench line
OUT=gpurun_out/r3b
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 900 python -m pytest tests/test_mpm_hip.py -m gpu -q --tb=short -p no:cacheprovider -s > $OUT/pytest_mpm.log 2>&1
grep -v "^Particles\|^Total\|^Setting\|^Material" $OUT/pytest_mpm.log | tail -30
for rep in 1 2; do
 for bits in 64 32; do
  PIXIE_MPM_WARM=400 PIXIE_MPM_BITS=$bits timeout 300 python scripts/mpm_bench.py 1000000 120 2000 2>&1 | grep "^n=" >> $OUT/variants.log
 done
done
for bits in 64 32; do for wide in 0 1; do
  PIXIE_MPM_WARM=400 PIXIE_MPM_BITS=$bits PIXIE_MPM_WIDE=$wide timeout 300 python scripts/mpm_bench.py 100000 50 2000 2>&1 | grep "^n=" >> $OUT/variants.log
done; done
cat $OUT/variants.log
for bits in 64 32; do
 (cd /tmp && PIXIE_MPM_WARM=400 PIXIE_MPM_BITS=$bits timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof$bits -o mpm -- python $ROOT/scripts/mpm_bench.py 1000000 120 1000 > $ROOT/$OUT/prof_mpm_$bits.log 2>&1)
 DB=$(find $OUT/prof$bits -name "*.db" | head -1)
 [ -n "$DB" ] && python scripts/rocpd_stats.py $DB $OUT/mpm_1m_bits${bits}_kernel_stats.csv $OUT/mpm_1m_bits${bits}_kernel_stats_by_geometry.csv
 rm -rf $OUT/prof$bits
 grep "^n=" $OUT/prof_mpm_$bits.log; head -6 $OUT/mpm_1m_bits${bits}_kernel_stats.csv | cut -c1-160
done
timeout 1200 python -m pytest tests/test_unet_hip.py -m gpu -q --tb=short -p no:cacheprovider -s -k "256_cube_128 or graph_replay or handle" > $OUT/pytest_unet_new.log 2>&1
tail -40 $OUT/pytest_unet_new.log
timeout 1200 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?"; tail -3 $OUT/bench.err; head -c 6000 $OUT/bench.json
