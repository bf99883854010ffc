#!/bin/bash
# Round 5, session C: two bounded experiments.
#  (1) conv3d_f16x3_c64_fullres_dpp_kernel (PIXIE_CONV_DPPB=1): dx = 1, 2 B fragments by v_mov_b32_dpp wave_shl:1 instead of LDS reads.
#      Parity (the U-Net GPU tests with the switch on), alternating timing of the dominant layer, energy per launch.
#  (2) set_scalar "compensated_x": displacement / v / C of the north-star scene against the float64 fixture, and its cost.
# Plus: what the host gives a container (cgroup quota) and how the OpenMP oracle scales with threads.
OUT=gpurun_out/${1:-r5c}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
(echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>&1)"; echo "cfs_quota_us: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>&1)"; echo "nproc: $(nproc)"; python -c "import os; print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))"; lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node\(s\)") > $OUT/host_cpus.txt 2>&1
R=$OUT/conv_dppb_ab.txt
: > $R
for rep in 1 2 3; do
  for v in 0 1; do
    echo "== PIXIE_CONV_DPPB=$v (repetition $rep)" >> $R
    PIXIE_CONV_DPPB=$v PIXIE_CONV_NSHAPES=1 timeout 300 python scripts/conv_bench.py 20 2>/dev/null | grep cin >> $R
  done
done
PIXIE_CONV_DPPB=1 timeout 900 python -m pytest tests/test_unet_hip.py -m gpu -q --tb=short -x -p no:cacheprovider -k "128 or conv3d or full" > $OUT/pytest_unet_dppb.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_unet_dppb.log
PIXIE_CONV_DPPB=0 timeout 300 python scripts/conv_energy.py > $OUT/conv_energy_base.txt 2>&1
PIXIE_CONV_DPPB=1 timeout 300 python scripts/conv_energy.py > $OUT/conv_energy_dppb.txt 2>&1
for v in 0 1; do PIXIE_CONV_DPPB=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-mpm --no-exact-f32 --no-unet-256 --no-shipped-shape --no-cpu-baseline > $OUT/bench_unet_dppb$v.json 2> /dev/null; done
timeout 900 python scripts/comp_x_experiment.py > $OUT/comp_x_experiment.txt 2>&1
for th in 1 2 4 8 16 32 64 128; do
  OMP_NUM_THREADS=$th timeout 120 python -c "
import sys, time; sys.path.insert(0, '.')
from oracle.mpm_oracle import OracleMPM
from pixie_amd.synthetic import apply_scene, mpm_ball_scene
sc = mpm_ball_scene(100000, seed=0, n_grid=50); o = OracleMPM(100000, 50, 2.0, 'f32_omp'); o.load_initial_data(sc['x'], sc['vol'], sc['cov']); apply_scene(o, sc)
o.run(sc['dt'], 1); t0 = time.perf_counter(); o.run(sc['dt'], 10); dt = time.perf_counter() - t0
print('OMP_NUM_THREADS=$th: %.3e particle-steps/s' % (1e6 / dt))" >> $OUT/omp_scaling.txt 2>&1
done
cat $OUT/host_cpus.txt; cat $R | cut -c1-200; tail -3 $OUT/pytest_unet_dppb.log; grep -E "J per launch|ms per launch|power|launch" $OUT/conv_energy_base.txt | head -8; echo ---; grep -E "J per launch|ms per launch|power|launch" $OUT/conv_energy_dppb.txt | head -8
for v in 0 1; do python -c "
import json; d = json.load(open('$OUT/bench_unet_dppb$v.json')); print('DPPB=$v ms_per_step', d['ms_per_step'], 'avg_launch_ms', d['roofline']['avg_launch_ms'], d.get('telemetry'))"; done
cat $OUT/comp_x_experiment.txt | grep -v "^Particles\|^Total\|^Setting\|^Material"; cat $OUT/omp_scaling.txt
