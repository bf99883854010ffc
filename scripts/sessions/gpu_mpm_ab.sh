#!/bin/bash
# A/B grid for the MPM block kernel.  Usage: gpu_mpm_ab.sh TAG
TAG=${1:-mpmab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
run () { env "$@" timeout 300 python scripts/mpm_bench.py $N $NG $ST 2>/dev/null | grep "^n=" | sed 's/particle-steps.*| fused/| fused/' | tee -a $OUT/ab.txt; }
N=1000000; NG=120; ST=300
for cap in 256 384 512 768 1024; do run PIXIE_MPM_ITEM_CAP=$cap; done
run PIXIE_MPM_ITEM_CAP=512 PIXIE_MPM_OCC=6
N=100000; NG=50; ST=1000
for cap in 256 384 512 1024; do run PIXIE_MPM_ITEM_CAP=$cap; done
timeout 600 python -m pytest tests/test_mpm_hip.py -m gpu -q --tb=short -p no:cacheprovider -x -k "phase_by_phase or rollout_parity or single_step" 2>&1 | tail -3
