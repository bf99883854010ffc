#!/bin/bash
# Round 6, session T: counters of the re-binning kernels on the sand configuration (where a re-binning every ~44 substeps is 8 % of the run).
OUT=gpurun_out/${1:-r6t}
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
  if [ -n "$f" ]; then python scripts/pmc_summary.py $f "bin_" > $OUT/pmc_$name.txt; else echo "no counter csv for $name" > $OUT/pmc_$name.txt; fi
  rm -rf $OUT/pmc_$name $OUT/pmc_$name.log
}
CMD="env PIXIE_MPM_SCENARIO=sand PIXIE_MPM_WARM=200 python $ROOT/scripts/mpm_bench.py 1000000 0 200"
run_pmc sq SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -- $CMD
run_pmc sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -- $CMD
run_pmc fetch FETCH_SIZE -- $CMD
run_pmc write WRITE_SIZE -- $CMD
run_pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum -- $CMD
cat $OUT/pmc_*.txt | cut -c1-60,100-700
