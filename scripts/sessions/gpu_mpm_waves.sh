#!/bin/bash
# sweep the occupancy target of the fused MPM kernel (rebuilds libpixie_hip.so on the box per setting)
TAG=${1:-r1i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics"
for w in 2 3 4 5; do
  hipcc $FLAGS -DPX_MPM_WAVES=$w -c pixie_amd/csrc/mpm.hip -o pixie_amd/build/mpm.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o pixie_amd/libpixie_hip.so pixie_amd/build/common.o pixie_amd/build/mpm.o pixie_amd/build/unet_ops.o pixie_amd/build/conv3d_mfma.o pixie_amd/build/conv3d_f16x3.o
  for n in "100000 50 1000" "1000000 120 300"; do
    echo -n "waves=$w " >> $OUT/waves.log
    timeout 200 python scripts/mpm_bench.py $n 32 2>&1 | grep "^n=" >> $OUT/waves.log
  done
done
hipcc $FLAGS -c pixie_amd/csrc/mpm.hip -o pixie_amd/build/mpm.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o pixie_amd/libpixie_hip.so pixie_amd/build/common.o pixie_amd/build/mpm.o pixie_amd/build/unet_ops.o pixie_amd/build/conv3d_mfma.o pixie_amd/build/conv3d_f16x3.o
timeout 600 python -m pytest tests/test_mpm_hip.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_mpm.log 2>&1
tail -3 $OUT/pytest_mpm.log
cat $OUT/waves.log
