#!/bin/bash
# round 4, session e: where the block kernel's time goes after the arithmetic regrouping (1M particles)
OUT=gpurun_out/r4e
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$(pwd)
B="python scripts/mpm_bench.py 1000000 120 1000"
run () { echo "== $1"; shift; env "$@" PIXIE_MPM_WARM=200 timeout 200 $B 2>&1 | grep "^n=" | cut -c1-330; }
{
run "default (5 waves/SIMD)"
run "occupancy 6" PIXIE_MPM_OCC=6
run "trace kernel, no ablation" PIXIE_MPM_TRACE=0x1000
run "trace kernel, LDS scatter atomics skipped" PIXIE_MPM_TRACE=0x1100
run "trace kernel, tile staging loads skipped" PIXIE_MPM_TRACE=0x1400
run "trace kernel, tile publish skipped" PIXIE_MPM_TRACE=0x1800
run "trace kernel, atomics + staging + publish skipped" PIXIE_MPM_TRACE=0x1d00
} > $OUT/ablations.txt 2>&1
cat $OUT/ablations.txt
python -m pytest tests/test_filling_hip.py -m gpu -q -p no:cacheprovider -k smoothing 2>&1 | tail -2
run_pmc () {  # name, counters..., -- cmd
  local name=$1; shift
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  (cd /tmp && timeout 400 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$name -o $name -- "$@" > $ROOT/$OUT/pmc_$name.log 2>&1)
  local f=$(find $OUT/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python scripts/pmc_summary.py $f > $OUT/pmc_$name.txt; else echo "no counter csv for $name" > $OUT/pmc_$name.txt; tail -5 $OUT/pmc_$name.log >> $OUT/pmc_$name.txt; fi
  rm -rf $OUT/pmc_$name
}
MPM="python $ROOT/scripts/mpm_bench.py 1000000 120 40 32"
run_pmc mpm_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -- $MPM
run_pmc mpm_sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE -- $MPM
for f in $OUT/pmc_mpm_sq.txt $OUT/pmc_mpm_sq2.txt; do grep -E "mpm_block_kernel<true, true|grid_block" $f | cut -c1-500; done
