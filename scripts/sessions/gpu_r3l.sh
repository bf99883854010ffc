#!/bin/bash
# LDS tile pitch variants of the fused MPM kernel: bit-identity test, A/B timing at 1 M and 100 k, LDS conflict counters
OUT=gpurun_out/r3l
mkdir -p $OUT
ROOT=$PWD
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_mpm_hip.py -m gpu -q --tb=short -p no:cacheprovider -k "pitch or packed or batched" > $OUT/pytest_sel.log 2>&1
grep -E "passed|failed|^E  " $OUT/pytest_sel.log | tail -5
for pad in 0 1 2 3; do
  PIXIE_MPM_PAD=$pad PIXIE_MPM_WARM=400 timeout 300 python scripts/mpm_bench.py 1000000 120 2000 2>&1 | grep "^n=" | tee -a $OUT/mpm_pad.txt
done
for pad in 0 1; do
  PIXIE_MPM_BITS=64 PIXIE_MPM_PAD=$pad PIXIE_MPM_WARM=400 timeout 300 python scripts/mpm_bench.py 1000000 120 2000 2>&1 | grep "^n=" | tee -a $OUT/mpm_pad.txt
done
for pad in 0 1 3; do
  PIXIE_MPM_PAD=$pad PIXIE_MPM_WARM=400 timeout 300 python scripts/mpm_bench.py 100000 64 4000 2>&1 | grep "^n=" | tee -a $OUT/mpm_pad.txt
done
for pad in 0 3; do
  (cd /tmp && PIXIE_MPM_PAD=$pad PIXIE_MPM_WARM=100 timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_pad$pad -o p -- python $ROOT/scripts/mpm_bench.py 1000000 120 200 > $ROOT/$OUT/pmc_pad$pad.log 2>&1)
  f=$(find $OUT/pmc_pad$pad -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python scripts/pmc_summary.py $f | grep -E "mpm_block|kernel" | head -8 > $OUT/pmc_pad$pad.txt; else tail -5 $OUT/pmc_pad$pad.log > $OUT/pmc_pad$pad.txt; fi
  rm -rf $OUT/pmc_pad$pad $OUT/pmc_pad$pad.log
  cat $OUT/pmc_pad$pad.txt
done
