#!/bin/bash
# Round-4 closing session on the final tree: the whole GPU suite (with the tests' own prints), smoke, the driver's bench command,
# the same command under rocprofv3 --stats, per-scene stats of the MPM step loop, FETCH/WRITE PMC passes of the MPM kernels at 1 M
# (the block kernel's arithmetic changed this round) with the calibration kernels in the same passes.
OUT=gpurun_out/${1:-r4fin}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$(pwd)
(rocm-smi --showproductname; rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -8) > $OUT/device.txt 2>&1
timeout 1700 python -m pytest tests -m gpu -q --tb=short -rA -s --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
grep -E "^(PASSED|FAILED|ERROR|SKIPPED)|passed|failed" $OUT/pytest_gpu.log | tail -400 > $OUT/pytest_gpu_tail.txt
grep -hE "oracle block on the recorded input|256\^3 x 128|bc test v|light-side|frame export vs|packed scatter, one substep|config 3|hip-vs-f64" $OUT/pytest_gpu.log | cut -c1-400 > $OUT/pytest_gpu_numbers.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" >> $OUT/smoke.log
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" >> $OUT/bench.err
cp gpurun_out/bench_detail.json $OUT/bench_detail.json 2>/dev/null
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --mpm-substeps 300 --mpm-large-substeps 300 --no-unet-256 > $ROOT/$OUT/prof_bench.json 2> $ROOT/$OUT/prof_bench.err)
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python scripts/rocpd_stats.py $DB $OUT/kernel_stats.csv $OUT/kernel_stats_by_geometry.csv
rm -rf $OUT/prof
for sc in "1000000 120 900 1m" "100000 50 2200 100k"; do
  set -- $sc
  (cd /tmp && PIXIE_MPM_WARM=100 timeout 400 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_$4 -o mpm -- python $ROOT/scripts/mpm_bench.py $1 $2 $3 > $ROOT/$OUT/mpm_$4_run.txt 2>&1)
  DB=$(find $OUT/prof_$4 -name "*.db" | head -1)
  [ -n "$DB" ] && python scripts/rocpd_stats.py $DB $OUT/mpm_$4_kernel_stats.csv /dev/null
  rm -rf $OUT/prof_$4
done
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
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2; tail -2 $OUT/smoke.log; tail -2 $OUT/bench.err; wc -c $OUT/bench.json; head -c 900 $OUT/bench.json; echo
head -6 $OUT/kernel_stats.csv | cut -c1-160; head -4 $OUT/mpm_1m_kernel_stats.csv | cut -c1-160; head -4 $OUT/mpm_100k_kernel_stats.csv | cut -c1-160
grep -A10 '"mpm_1m' $OUT/pmc_traffic.json | head -30
