#!/bin/bash
# Round 5, session F: the software-pipelined scene batch (pixie_amd/pipeline.py: neural_scene_batch) -- parity test + the bench's pipeline leg;
# run_batch / pack-kernel tests; H2D of the feature grid and the pack kernel timed in the bench.
OUT=gpurun_out/${1:-r5f}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_pipeline_hip.py tests/test_distributed.py tests/test_mpm_hip.py -m gpu -q --tb=short -rA -s -p no:cacheprovider -k "pipeline or pack or batch or stream or pipelined or device_resident" > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
timeout 900 python bench.py --steps 5 --warmup 2 --no-exact-f32 --no-unet-256 --no-shipped-shape --no-cpu-baseline --no-mpm-plastic --no-mpm-large > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" >> $OUT/bench.err
cp gpurun_out/bench_detail.json $OUT/bench_detail.json 2>/dev/null
grep -E "passed|failed|PASSED|FAILED|ERROR" $OUT/pytest.log | tail -15; tail -3 $OUT/bench.err
python -c "
import json; d = json.load(open('$OUT/bench.json'))
for k in ('ms_per_step', 'pipeline_ms_per_scene', 'pipeline_batch_ms_per_scene', 'pipeline_parts_ms', 'pack_fields_us', 'h2d_feature_grid_ms_pinned', 'mpm_us_per_substep', 'mpm_3_scenes_particle_steps_per_s'): print(k, d.get(k))"
