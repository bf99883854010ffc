#!/bin/bash
# Round-6 closing session on the final tree: COLD build of both libraries, the whole GPU suite, smoke, the reference's drivers (A-E), the
# driver's bench command (default legs) and `bench.py --full`, the default command under rocprofv3 --stats, per-scene kernel stats and SQ
# counters of the MPM block kernel, and FRESH FETCH/WRITE PMC passes of EVERY kernel the bench line quotes traffic for (conv 64->64 at 128^3,
# MPM 100 k, MPM 1 M) with the calibration kernels in the same passes; the f16 MFMA tap-loop microbench.
OUT=gpurun_out/${1:-r6last}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$(pwd)
(rocm-smi --showproductname; rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -8; echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>&1)"; nproc) > $OUT/device.txt 2>&1
PIXIE_FORCE_BUILD=1 timeout 600 python -c "import __graft_entry__ as g; g.build()" > $OUT/cold_build.log 2>&1
echo "build exit $?" >> $OUT/cold_build.log
timeout 1700 python -m pytest tests -m gpu -q --tb=short -rA -s --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
grep -E "^(PASSED|FAILED|ERROR|SKIPPED)|passed|failed" $OUT/pytest_gpu.log | tail -400 > $OUT/pytest_gpu_tail.txt
grep -hE "density vs the reference|oracle block on the recorded input|256\^3 x 128|bc test v|light-side|frame export vs|packed scatter, one substep|config 3|hip-vs-f64|@ substep|inverted particles|^(sand|metal|snow|visplas|water|mixed):|regrid:|compensated x|^(sand_rollout|metal_rollout|tree_rollout|jelly_apic|jelly_rpic|jelly_pic|mixed_materials|rotation_release|inverted) (64|32)" $OUT/pytest_gpu.log | cut -c1-1200 > $OUT/pytest_gpu_numbers.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" >> $OUT/smoke.log
timeout 900 python scripts/reference_drivers/run.py --only abcde --world 2 > $OUT/reference_drivers_stdout.log 2>&1
echo "drivers exit $?" >> $OUT/reference_drivers_stdout.log
cp gpurun_out/reference_drivers.log $OUT/reference_drivers.log 2>/dev/null
# counters first (bench.py reads profiles/pmc_traffic.json and profiles/mpm_counters.json of THIS tree: stamped with the kernels' source hashes)
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
scene_cmd () { if [ $1 = jelly ]; then echo "env PIXIE_MPM_WARM=100 python $ROOT/scripts/mpm_bench.py 1000000 120 $2"; elif [ $1 = jelly100k ]; then echo "env PIXIE_MPM_WARM=100 python $ROOT/scripts/mpm_bench.py 100000 50 $(( $2 * 6 ))"; else echo "env PIXIE_MPM_SCENARIO=$1 PIXIE_MPM_WARM=100 python $ROOT/scripts/mpm_bench.py 1000000 0 $2"; fi; }
for sc in jelly sand snow metal mixed jelly100k; do
  run_pmc sq_$sc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU -- $(scene_cmd $sc 40)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_$sc -o mpm -- $(scene_cmd $sc 300) > $ROOT/$OUT/run_$sc.txt 2>&1)
  DB=$(find $OUT/prof_$sc -name "*.db" | head -1)
  [ -n "$DB" ] && python scripts/rocpd_stats.py $DB $OUT/stats_$sc.csv
  rm -rf $OUT/prof_$sc
done
python scripts/mpm_counters.py $OUT $OUT/mpm_counters.json r6last > /dev/null
CAL="$ROOT/scripts/microbench/hbm_calib.exe"
CONV="python $ROOT/scripts/conv_bench.py 2"
M1M="python $ROOT/scripts/mpm_bench.py 1000000 120 60"
M100K="python $ROOT/scripts/mpm_bench.py 100000 50 300"
run_pmc calib_fetch FETCH_SIZE -- $CAL
run_pmc calib_write WRITE_SIZE -- $CAL
run_pmc conv_fetch FETCH_SIZE -- $CONV
run_pmc conv_write WRITE_SIZE -- $CONV
run_pmc mpm_1m_fetch FETCH_SIZE -- $M1M
run_pmc mpm_1m_write WRITE_SIZE -- $M1M
run_pmc mpm_100k_fetch FETCH_SIZE -- $M100K
run_pmc mpm_100k_write WRITE_SIZE -- $M100K
run_pmc conv_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT -- $CONV
python scripts/pmc_traffic.py $OUT $OUT/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json; cp $OUT/mpm_counters.json profiles/mpm_counters.json     # (on the box: what the bench runs below read)
timeout 120 scripts/microbench/mfma_lds.exe > $OUT/mfma_lds_microbench.txt 2>&1
timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" >> $OUT/bench.err
cp gpurun_out/bench_detail.json $OUT/bench_detail.json 2>/dev/null
timeout 2400 python bench.py --steps 20 --warmup 5 --full > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "bench --full exit $?" >> $OUT/bench_full.err
cp gpurun_out/bench_detail.json $OUT/bench_detail_full.json 2>/dev/null
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --mpm-substeps 300 --mpm-large-substeps 300 --mpm-plastic-substeps 100 > $ROOT/$OUT/prof_bench.json 2> $ROOT/$OUT/prof_bench.err)
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python scripts/rocpd_stats.py $DB $OUT/kernel_stats.csv $OUT/kernel_stats_by_geometry.csv
rm -rf $OUT/prof
tail -4 $OUT/cold_build.log; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu_tail.txt | head; tail -2 $OUT/smoke.log; tail -4 $OUT/reference_drivers_stdout.log | cut -c1-300
tail -2 $OUT/bench.err; wc -c $OUT/bench.json; cat $OUT/bench.json; echo; tail -2 $OUT/bench_full.err; wc -c $OUT/bench_full.json
head -6 $OUT/kernel_stats.csv | cut -c1-160; cat $OUT/mpm_counters.json | grep -E "\"1m_|\"100k|valu_per_wave|block_kernel_us|grid_kernel_us"; grep -E '"(conv|mpm)_|hbm_bytes' $OUT/pmc_traffic.json | head -24
