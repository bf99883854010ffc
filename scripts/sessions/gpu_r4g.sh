#!/bin/bash
# round 4, session g: floors and energy -- dependent launch pairs, work-item timelines of the new block kernel, energy per
# conv launch vs the bare MFMA loop, then the driver's bench command
OUT=gpurun_out/r4g
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 120 scripts/microbench/launch_chain.exe 2>&1 | tee $OUT/launch_chain.txt
timeout 200 python scripts/mpm_trace.py 100000 50 2>&1 | grep -v "^Particles\|^Total\|^Setting\|^Material\|amdgpu.ids" | tee $OUT/mpm_trace_100k.txt
timeout 200 python scripts/mpm_trace.py 1000000 120 2>&1 | grep -v "^Particles\|^Total\|^Setting\|^Material\|amdgpu.ids" | tee $OUT/mpm_trace_1m.txt
timeout 300 python scripts/conv_energy.py 2>&1 | grep -v "amdgpu.ids" | tee $OUT/conv_energy.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
wc -c $OUT/bench.json
cp gpurun_out/bench_detail.json $OUT/bench_detail.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4g/bench.json"))
for k in ("value", "ms_per_step", "mpm_us_per_substep", "mpm_frac_dense", "mpm_frac_touched", "mpm_1m_us_per_substep", "mpm_1m_frac_dense", "mpm_1m_frac_touched",
          "exact_f32_voxels_per_s", "p2g2p_loop_vs_run", "mpm_3_scenes_particle_steps_per_s", "mpm_6_scenes_particle_steps_per_s", "unet_256x128_ms_per_step"):
    print(k, d.get(k))
print("roofline", d.get("roofline")); print("cpu", d.get("cpu_baseline")); print("mpm cpu", d.get("mpm_cpu_baseline")); print("mpm_1m cpu", d.get("mpm_1m_cpu_baseline"))
PY
