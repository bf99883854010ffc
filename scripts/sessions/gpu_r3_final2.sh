#!/bin/bash
# Round-3 closing session on the final tree (PMC passes: profiles/r3end_pmc_*, kernels unchanged since): the whole GPU suite,
# smoke, the driver's bench command, the same command under rocprofv3 --stats.
OUT=gpurun_out/${1:-r3fin}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$(pwd)
(rocm-smi --showproductname; rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -8) > $OUT/device.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q --tb=short -rA --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" >> $OUT/smoke.log
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" >> $OUT/bench.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --mpm-substeps 300 --mpm-large-substeps 300 --no-unet-256 > $ROOT/$OUT/prof_bench.json 2> $ROOT/$OUT/prof_bench.err)
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python scripts/rocpd_stats.py $DB $OUT/kernel_stats.csv $OUT/kernel_stats_by_geometry.csv
rm -rf $OUT/prof
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2; tail -2 $OUT/smoke.log; tail -2 $OUT/bench.err; head -c 600 $OUT/bench.json; echo; head -8 $OUT/kernel_stats.csv | cut -c1-150
