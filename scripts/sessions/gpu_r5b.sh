#!/bin/bash
# Round 5, session B: the whole GPU suite on the two-library tree (new: pipeline, pack kernel, stream ordering), smoke, the reference's
# drivers incl. the sharded U-Net program (part E; the GPU exposed twice if the runtime allows it), the driver's bench command,
# the f16 MFMA tap-loop microbench re-measured on this box, jelly at sand's n_grid for the equal-size ratio.
OUT=gpurun_out/${1:-r5b}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$(pwd)
(rocm-smi --showproductname; rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -8) > $OUT/device.txt 2>&1
HIP_VISIBLE_DEVICES=0,0 timeout 120 python -c "import torch; print('HIP_VISIBLE_DEVICES=0,0 -> device_count', torch.cuda.device_count())" > $OUT/two_devices_probe.txt 2>&1
ROCR_VISIBLE_DEVICES=0,0 timeout 120 python -c "import torch; print('ROCR_VISIBLE_DEVICES=0,0 -> device_count', torch.cuda.device_count())" >> $OUT/two_devices_probe.txt 2>&1
timeout 1700 python -m pytest tests -m gpu -q --tb=short -rA -s --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
grep -E "^(PASSED|FAILED|ERROR|SKIPPED)|passed|failed" $OUT/pytest_gpu.log | tail -400 > $OUT/pytest_gpu_tail.txt
grep -hE "density vs the reference|oracle block on the recorded input|256\^3 x 128|bc test v|light-side|frame export vs|packed scatter, one substep|config 3|hip-vs-f64|@ substep|inverted particles|^(sand|metal|snow|visplas|water):|regrid:" $OUT/pytest_gpu.log | cut -c1-400 > $OUT/pytest_gpu_numbers.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" >> $OUT/smoke.log
NDEV=1; grep -q "device_count 2" $OUT/two_devices_probe.txt && NDEV=2
if grep -q "HIP_VISIBLE_DEVICES=0,0 -> device_count 2" $OUT/two_devices_probe.txt; then export_two="HIP_VISIBLE_DEVICES=0,0"; elif grep -q "ROCR_VISIBLE_DEVICES=0,0 -> device_count 2" $OUT/two_devices_probe.txt; then export_two="ROCR_VISIBLE_DEVICES=0,0"; else export_two="PIXIE_NO_SECOND_DEVICE=1"; fi
env $export_two timeout 900 python scripts/reference_drivers/run.py --only e --world 2 > $OUT/reference_drivers_e_stdout.log 2>&1
echo "drivers(e) exit $?" >> $OUT/reference_drivers_e_stdout.log
cp gpurun_out/reference_drivers.log $OUT/reference_drivers_e.log 2>/dev/null
timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" >> $OUT/bench.err
cp gpurun_out/bench_detail.json $OUT/bench_detail.json 2>/dev/null
timeout 120 scripts/microbench/mfma_lds.exe > $OUT/mfma_lds_microbench.txt 2>&1
(PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 200 400 2>&1 | grep "us/substep" | cut -c1-330) > $OUT/jelly_ngrid200.txt
cat $OUT/two_devices_probe.txt; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu_tail.txt | head; tail -2 $OUT/smoke.log
tail -6 $OUT/reference_drivers_e_stdout.log | cut -c1-400; tail -3 $OUT/bench.err; wc -c $OUT/bench.json; cat $OUT/bench.json; echo; cat $OUT/jelly_ngrid200.txt; grep RANDOM $OUT/mfma_lds_microbench.txt
