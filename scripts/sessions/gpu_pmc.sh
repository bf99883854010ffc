#!/bin/bash
# PMC passes (counters only, with --kernel-trace) for the MPM block kernel and the dominant conv. -> gpurun_out/$TAG
TAG=${1:-r1k}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
(cd /tmp && rocprofv3 -L > $ROOT/$OUT/counters_list.txt 2>&1)
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
run_pmc mpm_fetch FETCH_SIZE -- $MPM
run_pmc mpm_write WRITE_SIZE -- $MPM
CONV="python $ROOT/scripts/conv_bench.py 2"
run_pmc conv_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT -- $CONV
run_pmc conv_sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE SQ_WAVES -- $CONV
run_pmc conv_fetch FETCH_SIZE -- $CONV
run_pmc conv_write WRITE_SIZE -- $CONV
grep -c . $OUT/counters_list.txt
for f in $OUT/pmc_*.txt; do echo "== $f"; grep -E "mpm_block_kernel<true, true>|grid_block|f16x3_kernel<3, 2, 4>|no counter|rror" $f | cut -c1-600; done
