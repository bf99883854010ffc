#!/bin/bash
# Round 6, session A: the one-launch substep (block kernel consumes the previous P2G's tiles) -- parity, then timing at 100 k and 1 M.
OUT=gpurun_out/${1:-r6a}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_mpm_hip.py -m gpu -q --tb=short -p no:cacheprovider -k "one_launch or sparse_tile or latency_optimised or slow_path" -s > $OUT/pytest_one.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_one.log
R=$OUT/one_launch_timing.txt
: > $R
for rep in 1 2; do
  for one in 0 1; do
    (PIXIE_MPM_ONE=$one PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 100000 50 3000 2>&1 | grep "us/substep" | cut -c1-330) >> $R
  done
done
for one in 0 1; do
  (PIXIE_MPM_ONE=$one PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 120 400 2>&1 | grep "us/substep" | cut -c1-330) >> $R
  (PIXIE_MPM_ONE=$one PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 300000 120 600 2>&1 | grep "us/substep" | cut -c1-330) >> $R
  (PIXIE_MPM_ONE=$one PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 2000 50 3000 2>&1 | grep "us/substep" | cut -c1-330) >> $R
  (PIXIE_MPM_ONE=$one PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 20000 50 3000 2>&1 | grep "us/substep" | cut -c1-330) >> $R
done
tail -15 $OUT/pytest_one.log; cat $R | cut -c1-250
