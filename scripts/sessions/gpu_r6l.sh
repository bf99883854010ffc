export TMPDIR=/tmp
cp pixie_amd/libpixie_hip.so /tmp/new.so
for rep in 1 2; do
for which in shipped max-ilp max-memory-clause; do
  if [ $which = shipped ]; then cp /tmp/new.so pixie_amd/libpixie_hip.so; else cp scripts/_ab/libpixie_hip_$which.so pixie_amd/libpixie_hip.so; fi
  for args in "100000 50 3000" "1000000 120 400"; do
    PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py $args 2>&1 | grep "us/substep" | sed "s/^/$which /" | cut -c1-40,110-135
  done
  PIXIE_MPM_SCENARIO=metal PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 300 2>&1 | grep "us/substep" | sed "s/^/$which /" | cut -c1-40,110-135
done; done
cp /tmp/new.so pixie_amd/libpixie_hip.so
