#!/bin/bash
# Round-4 confirmation on the final tree (after the closing session r4fin): the reference's drivers incl. the particle pre-pass
# (part D), the whole GPU suite with the new reference-code filling fixture, smoke, and the driver's bench command.
OUT=gpurun_out/${1:-r4z}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python scripts/reference_drivers/run.py > $OUT/reference_drivers_stdout.log 2>&1
echo "drivers exit $?" >> $OUT/reference_drivers_stdout.log
cp gpurun_out/reference_drivers.log $OUT/reference_drivers.log 2>/dev/null
timeout 1700 python -m pytest tests -m gpu -q --tb=short -rA -s --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
grep -E "^(PASSED|FAILED|ERROR|SKIPPED)|passed|failed" $OUT/pytest_gpu.log | tail -400 > $OUT/pytest_gpu_tail.txt
grep -hE "density vs the reference|oracle block on the recorded input|256\^3 x 128|bc test v|light-side|frame export vs|packed scatter, one substep|config 3|hip-vs-f64" $OUT/pytest_gpu.log | cut -c1-400 > $OUT/pytest_gpu_numbers.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" >> $OUT/smoke.log
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" >> $OUT/bench.err
cp gpurun_out/bench_detail.json $OUT/bench_detail.json 2>/dev/null
tail -12 $OUT/reference_drivers_stdout.log | cut -c1-300
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2; grep FAILED $OUT/pytest_gpu_tail.txt | head; cat $OUT/pytest_gpu_numbers.txt | grep "density vs"; tail -2 $OUT/smoke.log; tail -2 $OUT/bench.err; wc -c $OUT/bench.json; head -c 1200 $OUT/bench.json; echo
