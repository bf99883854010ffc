#!/bin/bash
# packed-fp32 transfer cores, fixed vector literals: bit-identity test, A/B timing at 4 and 5 waves per SIMD
OUT=gpurun_out/r3n
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_mpm_hip.py -m gpu -q --tb=short -p no:cacheprovider -k "variants or packed or batched or phase" > $OUT/pytest_sel.log 2>&1
grep -E "passed|failed|^E  " $OUT/pytest_sel.log | tail -5
b () { env "$@" PIXIE_MPM_WARM=400 timeout 300 python scripts/mpm_bench.py $N $NG $STEPS 2>&1 | grep "^n=" | tee -a $OUT/mpm_pk.txt; }
N=1000000; NG=120; STEPS=2000
b PIXIE_MPM_PK=0 PIXIE_MPM_PAD=0
b PIXIE_MPM_PK=1 PIXIE_MPM_PAD=0
b PIXIE_MPM_PK=1 PIXIE_MPM_PAD=0 PIXIE_MPM_OCC=4
b PIXIE_MPM_PK=0 PIXIE_MPM_PAD=0 PIXIE_MPM_OCC=4
b PIXIE_MPM_PK=1 PIXIE_MPM_PAD=3 PIXIE_MPM_OCC=4
N=100000; NG=50; STEPS=4000
b PIXIE_MPM_PK=0 PIXIE_MPM_PAD=0
b PIXIE_MPM_PK=1 PIXIE_MPM_PAD=0
