#!/bin/bash
# Short confirmation of the committed tree on the GPU box: all GPU tests, build() + smoke() in ONE process, the default bench.
TAG=${1:-confirm}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1700 python -m pytest tests -m gpu -q --tb=short --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" >> $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" >> $OUT/bench.err
tail -3 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log; tail -1 $OUT/bench.err; head -c 600 $OUT/bench.json; echo
