#!/bin/bash
# round 4, session k: (1) the 8-wave one-workgroup-per-CU conv tile against the shipped 4-wave two-per-CU kernel; (2) which
# block-kernel variant scenes that share a GPU should run
OUT=gpurun_out/r4k
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for nw in 0 1; do
  echo "== PIXIE_CONV_NW8=$nw" | tee -a $OUT/conv_nw8.txt
  if [ $nw = 1 ]; then export PIXIE_CONV_NW8=1; else unset PIXIE_CONV_NW8; fi
  PIXIE_CONV_NSHAPES=3 timeout 300 python scripts/conv_bench.py 10 2>&1 | grep cin | tee -a $OUT/conv_nw8.txt
done
export PIXIE_CONV_NW8=1
timeout 600 python -m pytest tests/test_unet_hip.py -m gpu -q -p no:cacheprovider -k "north_star or prologue_variants or epilogue_statistics" 2>&1 | tail -3 | tee -a $OUT/conv_nw8.txt
timeout 300 python bench.py --steps 5 --warmup 2 --no-mpm --no-cpu-baseline --no-unet-256 --no-shipped-shape --no-exact-f32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NW8 bench ms_per_step', d['ms_per_step'], d['roofline']['avg_launch_ms'])" | tee -a $OUT/conv_nw8.txt
unset PIXIE_CONV_NW8
timeout 300 python bench.py --steps 5 --warmup 2 --no-mpm --no-cpu-baseline --no-unet-256 --no-shipped-shape --no-exact-f32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('shipped bench ms_per_step', d['ms_per_step'], d['roofline']['avg_launch_ms'])" | tee -a $OUT/conv_nw8.txt
for w in -1 0; do
  PIXIE_MPM_MULTI_WIDE=$w timeout 300 python scripts/mpm_multi_scene.py 100000 50 2000 3,6 2>&1 | grep "scenes x" | sed "s/^/multi_wide=$w /" | tee -a $OUT/multi.txt
done
