#!/bin/bash
# Round 6, session E: crushed-frame refinement + rank-deficient rebuild in the block kernel, the RCCL wire path on one rank,
# timing sanity on every MPM scene, then the whole GPU suite.
OUT=gpurun_out/${1:-r6e}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$OUT/timing.txt
: > $R
(PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 120 400 2>&1 | grep "us/substep" | cut -c1-420) >> $R
(PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 100000 50 3000 2>&1 | grep "us/substep" | cut -c1-420) >> $R
for sc in sand snow metal mixed; do (PIXIE_MPM_SCENARIO=$sc PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 400 2>&1 | grep "us/substep" | cut -c1-420) >> $R; done
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
cut -c1-200 $R; tail -6 $OUT/pytest.log
