#!/bin/bash
# One GPU-box session: parity tests, bench, rocprofv3 kernel stats.  Everything lands in gpurun_out/.
# usage: scripts/sessions/gpu_session.sh [tag]
TAG=${1:-r1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
(rocm-smi --showproductname --showmeminfo vram; rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -12) > $OUT/device.txt 2>&1
python - > $OUT/build_check.txt 2>&1 <<'PY'
import __graft_entry__ as g
g.build()
print("build ok")
PY
timeout 1500 python -m pytest tests -m gpu -q --tb=short -rA --timeout=600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" >> $OUT/smoke.log
timeout 900 python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" >> $OUT/bench.err
ROOT=$(pwd)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --mpm-substeps 300 > $ROOT/$OUT/prof_bench.json 2> $ROOT/$OUT/prof_bench.err)
find $OUT/prof -name "*stats*" | head; ls -la $OUT/prof 2>/dev/null | head
tail -5 $OUT/pytest_gpu.log; cat $OUT/smoke.log | tail -3; cat $OUT/bench.json | head -c 3000
