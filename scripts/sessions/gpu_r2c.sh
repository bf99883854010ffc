#!/bin/bash
# Round-2 GPU session C: VALU issue-rate microbenchmark, PMC pass on the block kernel, new odd-grid / export tests.
TAG=${1:-r2c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$(pwd)
./scripts/microbench/valu_rate.exe > $OUT/valu_rate.txt 2>&1
run_pmc () {  # name, counters..., -- cmd
  local name=$1; shift
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  (cd /tmp && timeout 400 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$name -o $name -- "$@" > $ROOT/$OUT/pmc_$name.log 2>&1)
  local f=$(find $OUT/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python scripts/pmc_summary.py $f | grep -E "mpm_block_kernel<true, true|grid_block" > $OUT/pmc_$name.txt; else echo "no counter csv for $name" > $OUT/pmc_$name.txt; tail -5 $OUT/pmc_$name.log >> $OUT/pmc_$name.txt; fi
  rm -rf $OUT/pmc_$name $OUT/pmc_$name.log
}
M1M="python $ROOT/scripts/mpm_bench.py 1000000 120 60"
run_pmc mpm_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES -- $M1M
PIXIE_MPM_TRACE=0xf00 run_pmc mpm_sq_ablated SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES -- $M1M
run_pmc mpm_sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- $M1M
timeout 900 python -m pytest tests/test_unet_hip.py tests/test_mpm_hip.py tests/test_field_mapping_hip.py -m gpu -q --tb=short -p no:cacheprovider -k "golden or exports or mapping or voxel_points or single_step" > $OUT/pytest_subset.log 2>&1
cat $OUT/valu_rate.txt; cat $OUT/pmc_*.txt | cut -c1-700; tail -5 $OUT/pytest_subset.log
