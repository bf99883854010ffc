#!/bin/bash
# Round 6, session O: what bounds the block kernel of the sparse sand scene (75 us for the same particle count and VALU work as jelly's 51 us)?  Counters of both.
OUT=gpurun_out/${1:-r6o}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$(pwd)
run_pmc () {
  local name=$1; shift
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  (cd /tmp && timeout 300 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$name -o $name -- "$@" > $ROOT/$OUT/pmc_$name.log 2>&1)
  local f=$(find $OUT/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python scripts/pmc_summary.py $f "mpm_block_kernel<true, true" > $OUT/pmc_$name.txt; else echo "no counter csv for $name" > $OUT/pmc_$name.txt; fi
  rm -rf $OUT/pmc_$name $OUT/pmc_$name.log
}
for sc in sand jelly; do
  if [ $sc = sand ]; then CMD="env PIXIE_MPM_SCENARIO=sand PIXIE_MPM_WARM=100 python $ROOT/scripts/mpm_bench.py 1000000 0 40"; else CMD="env PIXIE_MPM_WARM=100 python $ROOT/scripts/mpm_bench.py 1000000 120 40"; fi
  run_pmc sq_$sc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -- $CMD
  run_pmc sq2_$sc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -- $CMD
  run_pmc fetch_$sc FETCH_SIZE -- $CMD
  run_pmc write_$sc WRITE_SIZE -- $CMD
  run_pmc tcc_$sc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -- $CMD
done
grep -h "5, 2>" $OUT/pmc_*.txt | grep "dispatches [0-9][0-9]" | cut -c1-30,100-600
