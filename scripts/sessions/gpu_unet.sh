#!/bin/bash
# U-Net iteration loop on the GPU box.  Usage: gpu_unet.sh TAG [pytest -k expression]
TAG=${1:-unet}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_unet_hip.py -m gpu -q --tb=short -p no:cacheprovider -x -s -k "${2:-fused or graph or golden}" > $OUT/pytest_unet.log 2>&1
grep -E "passed|failed|Error|error|fused grid|forward: eager" $OUT/pytest_unet.log | tail -12
grep -E "^E  " $OUT/pytest_unet.log | head -10
timeout 600 python bench.py --steps 2 --warmup 1 --no-mpm --no-cpu-baseline --no-exact-f32 > $OUT/bench_unet.json 2> $OUT/bench_unet.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_unet.json").read().strip().splitlines()[-1])
print("128^3:", d["ms_per_step"], "ms/scene"); print(json.dumps(d.get("shipped_shape_64x768"), indent=1))
PY
tail -3 $OUT/bench_unet.err
