#!/bin/bash
# Full GPU session for a round: microbenchmarks (raw output), PMC traffic passes -> profiles/pmc_traffic.json, parity tests,
# smoke, bench, rocprofv3 kernel stats of the same bench command.  Usage: gpu_round.sh TAG
TAG=${1:-r2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$(pwd)
(rocm-smi --showproductname; rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -8) > $OUT/device.txt 2>&1
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > $OUT/build_check.txt 2>&1
for mb in mfma_lds valu_rate valu_rate2 scatter_variants lds_atomics; do
  [ -x scripts/microbench/$mb.exe ] && timeout 120 ./scripts/microbench/$mb.exe > $OUT/${mb}_microbench.txt 2>&1
done
run_pmc () {  # name, counters..., -- cmd
  local name=$1; shift
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  (cd /tmp && timeout 400 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$name -o $name -- "$@" > $ROOT/$OUT/pmc_$name.log 2>&1)
  local f=$(find $OUT/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python scripts/pmc_summary.py $f > $OUT/pmc_$name.txt; else echo "no counter csv for $name" > $OUT/pmc_$name.txt; tail -5 $OUT/pmc_$name.log >> $OUT/pmc_$name.txt; fi
  rm -rf $OUT/pmc_$name $OUT/pmc_$name.log
}
CAL="$ROOT/scripts/microbench/hbm_calib.exe"
M100="python $ROOT/scripts/mpm_bench.py 100000 50 200"
M1M="python $ROOT/scripts/mpm_bench.py 1000000 120 60"
export PIXIE_CONV_NSHAPES=1
CONV="python $ROOT/scripts/conv_bench.py 3"
run_pmc calib_fetch FETCH_SIZE -- $CAL
run_pmc calib_write WRITE_SIZE -- $CAL
run_pmc conv_fetch FETCH_SIZE -- $CONV
run_pmc conv_write WRITE_SIZE -- $CONV
run_pmc mpm_100k_fetch FETCH_SIZE -- $M100
run_pmc mpm_100k_write WRITE_SIZE -- $M100
run_pmc mpm_1m_fetch FETCH_SIZE -- $M1M
run_pmc mpm_1m_write WRITE_SIZE -- $M1M
run_pmc conv_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT -- $CONV
run_pmc mpm_1m_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES -- $M1M
run_pmc mpm_1m_sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -- $M1M
unset PIXIE_CONV_NSHAPES
python scripts/pmc_traffic.py $OUT $OUT/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
cp $OUT/mfma_lds_microbench.txt profiles/${TAG}_mfma_lds_microbench.txt 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q --tb=short -rA --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" >> $OUT/smoke.log
timeout 1200 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" >> $OUT/bench.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-f32 --mpm-substeps 300 > $ROOT/$OUT/prof_bench.json 2> $ROOT/$OUT/prof_bench.err)
timeout 300 python scripts/unet_exec_bench.py 16 32 64 128 2>/dev/null | grep "D=" > $OUT/unet_exec_bench.txt
bash scripts/sessions/gpu_conv_wino.sh $TAG > /dev/null 2>&1
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python scripts/rocpd_stats.py $DB $OUT/kernel_stats.csv
rm -rf $OUT/prof
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2; tail -2 $OUT/smoke.log; tail -3 $OUT/bench.err; head -c 3000 $OUT/bench.json; echo; head -14 $OUT/kernel_stats.csv | cut -c1-150
