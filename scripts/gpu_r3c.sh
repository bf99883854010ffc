#!/bin/bash
# round 3, session c: U-Net graph diagnosis at 256^3, conv start-stagger experiment, MPM two-mode tests
OUT=gpurun_out/r3c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python scripts/unet_diag.py 256 128 cont > $OUT/unet_diag.txt 2>&1
timeout 200 python scripts/unet_diag.py 128 64 cont >> $OUT/unet_diag.txt 2>&1
grep "^D=" $OUT/unet_diag.txt
export PIXIE_CONV_NSHAPES=3
for st in 0 4 8 12 16 20 28 40; do
  echo "== start stagger $st us (conv_dbg $((st * 256)))"
  PIXIE_CONV_DBG=$((st * 256)) timeout 300 python scripts/conv_bench.py 8 2>/dev/null | grep cin | sed -e 's/f32 .*TF)  f16x3/f16x3/' -e 's/speedup.*//'
done | tee $OUT/conv_start_stagger.txt
unset PIXIE_CONV_NSHAPES
timeout 900 python -m pytest tests/test_mpm_hip.py -m gpu -q --tb=short -p no:cacheprovider -s > $OUT/pytest_mpm.log 2>&1
grep -v "^Particles\|^Total\|^Setting\|^Material" $OUT/pytest_mpm.log | tail -60
timeout 600 python -m pytest tests/test_unet_hip.py -m gpu -q --tb=short -p no:cacheprovider -s -k "graph_replay or fused_voxel or two_independent" > $OUT/pytest_unet_sel.log 2>&1
tail -12 $OUT/pytest_unet_sel.log
