#!/bin/bash
# Round 6, session B: evidence for the one-launch substep (rocprofv3 kernel durations, FETCH/WRITE, TCC hit/miss, VALU per wave; both
# modes, 100 k and 1 M), the cross-stream hop microbench, the grid kernel's loads-in-flight at 100 k.
OUT=gpurun_out/${1:-r6b}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$(pwd)
timeout 900 python -m pytest tests/test_mpm_hip.py -m gpu -q --tb=short -p no:cacheprovider -k "one_launch" -s > $OUT/pytest_one.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_one.log
timeout 120 scripts/microbench/stream_hop.exe > $OUT/stream_hop.txt 2>&1
run_pmc () {  # name, counters..., -- cmd
  local name=$1; shift
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  (cd /tmp && timeout 300 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$name -o $name -- "$@" > $ROOT/$OUT/pmc_$name.log 2>&1)
  local f=$(find $OUT/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python scripts/pmc_summary.py $f mpm_ > $OUT/pmc_$name.txt; else echo "no counter csv for $name" > $OUT/pmc_$name.txt; tail -5 $OUT/pmc_$name.log >> $OUT/pmc_$name.txt; fi
  rm -rf $OUT/pmc_$name $OUT/pmc_$name.log
}
for size in 100k 1m; do
  if [ $size = 100k ]; then ARGS="100000 50"; N=1500; NP=200; else ARGS="1000000 120"; N=300; NP=40; fi
  for one in 0 1; do
    T=${size}_one$one
    (cd /tmp && PIXIE_MPM_ONE=$one PIXIE_MPM_WARM=100 timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_$T -o mpm -- python $ROOT/scripts/mpm_bench.py $ARGS $N > $ROOT/$OUT/run_$T.txt 2>&1)
    DB=$(find $OUT/prof_$T -name "*.db" | head -1)
    [ -n "$DB" ] && python scripts/rocpd_stats.py $DB $OUT/stats_$T.csv
    rm -rf $OUT/prof_$T
    CMD="env PIXIE_MPM_ONE=$one PIXIE_MPM_WARM=100 python $ROOT/scripts/mpm_bench.py $ARGS $NP"
    run_pmc sq_$T SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -- $CMD
    run_pmc fetch_$T FETCH_SIZE -- $CMD
    run_pmc write_$T WRITE_SIZE -- $CMD
    run_pmc tcc_$T TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -- $CMD
  done
done
R=$OUT/grid_rb_100k.txt
: > $R
for rb in 0 2 4 1; do
  (PIXIE_MPM_GRID_RB=$rb PIXIE_MPM_DIAG=1 PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 100000 50 3000 2>&1 | grep "us/substep" | sed "s/^/grid_rb=$rb /" | cut -c1-420) >> $R
done
tail -4 $OUT/pytest_one.log; cat $OUT/stream_hop.txt; cat $R | cut -c1-60,180-420; for f in $OUT/stats_*.csv; do echo == $f; grep -E "mpm_(block|substep|grid_block)" $f | cut -c1-40,150-; done; grep -h "us/substep" $OUT/run_*.txt | cut -c1-200; cat $OUT/pmc_*.txt | cut -c1-300
