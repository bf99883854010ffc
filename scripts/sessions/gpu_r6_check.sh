#!/bin/bash
# Round 6: final-tree check after the closing session -- whole GPU suite, smoke, the driver's default bench command.
OUT=gpurun_out/${1:-r6check}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1700 python -m pytest tests -m gpu -q --tb=short --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" >> $OUT/smoke.log
( time timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench_time.txt
echo "bench exit $?" >> $OUT/bench.err
tail -3 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log; tail -3 $OUT/bench.err; cat $OUT/bench_time.txt; wc -c $OUT/bench.json; head -c 700 $OUT/bench.json
