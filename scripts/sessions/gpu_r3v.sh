#!/bin/bash
# HBM traffic of the MPM kernels at 1 M particles after the sparse tiles: FETCH_SIZE / WRITE_SIZE passes with the calibration
# kernels; the conv and 100 k passes are those of r3end (kernels unchanged).
OUT=gpurun_out/r3v
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$(pwd)
for f in conv_fetch conv_write mpm_100k_fetch mpm_100k_write; do cp profiles/r3end_pmc_$f.txt $OUT/pmc_$f.txt; done
run_pmc () {  # name, counters..., -- cmd
  local name=$1; shift
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  (cd /tmp && timeout 300 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$name -o $name -- "$@" > $ROOT/$OUT/pmc_$name.log 2>&1)
  local f=$(find $OUT/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python scripts/pmc_summary.py $f > $OUT/pmc_$name.txt; else echo "no counter csv for $name" > $OUT/pmc_$name.txt; tail -5 $OUT/pmc_$name.log >> $OUT/pmc_$name.txt; fi
  rm -rf $OUT/pmc_$name $OUT/pmc_$name.log
}
CAL="$ROOT/scripts/microbench/hbm_calib.exe"
M1M="python $ROOT/scripts/mpm_bench.py 1000000 120 60"
run_pmc calib_fetch FETCH_SIZE -- $CAL
run_pmc calib_write WRITE_SIZE -- $CAL
run_pmc mpm_1m_fetch FETCH_SIZE -- $M1M
run_pmc mpm_1m_write WRITE_SIZE -- $M1M
python scripts/pmc_traffic.py $OUT $OUT/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1
grep -A8 "mpm_1m" $OUT/pmc_traffic.json | head -40
