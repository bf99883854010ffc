#!/bin/bash
# Round 6, session Q: the lean gather of the grid kernel (816 -> ~300 VALU instructions per wave; no predicated loads) -- A/B against the commit
# before it (alternating libraries), state hashes must be equal; kernel durations by rocprofv3; MPM test files.
OUT=gpurun_out/${1:-r6q}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$(pwd)
R=$OUT/ab.txt
: > $R
cp pixie_amd/libpixie_hip.so /tmp/new.so; cp pixie_amd/libpixie_hip_diag.so /tmp/new_diag.so
for rep in 1 2 3; do
  for which in new prev; do
    if [ $which = new ]; then cp /tmp/new.so pixie_amd/libpixie_hip.so; else cp scripts/_ab/libpixie_hip_prev.so pixie_amd/libpixie_hip.so; fi
    (PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 120 400 2>&1 | grep "us/substep" | sed "s/^/$which /" | cut -c1-140) >> $R
    (PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 100000 50 3000 2>&1 | grep "us/substep" | sed "s/^/$which /" | cut -c1-140) >> $R
    (PIXIE_MPM_SCENARIO=sand PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 400 2>&1 | grep "us/substep" | sed "s/^/$which /" | cut -c1-140) >> $R
    (PIXIE_MPM_SCENARIO=metal PIXIE_MPM_WARM=100 timeout 200 python scripts/mpm_bench.py 1000000 0 300 2>&1 | grep "us/substep" | sed "s/^/$which /" | cut -c1-140) >> $R
  done
done
for which in new prev; do
  if [ $which = new ]; then cp /tmp/new.so pixie_amd/libpixie_hip.so; else cp scripts/_ab/libpixie_hip_prev.so pixie_amd/libpixie_hip.so; fi
  echo "== $which" >> $OUT/state_hash.txt
  timeout 200 python scripts/mpm_state_hash.py 100000 50 400 2>/dev/null | grep sha256 >> $OUT/state_hash.txt
  timeout 200 python scripts/mpm_state_hash.py 1000000 120 200 2>/dev/null | grep sha256 >> $OUT/state_hash.txt
  timeout 200 python scripts/mpm_state_hash.py 100000 50 400 64 2>/dev/null | grep sha256 >> $OUT/state_hash.txt
done
cp /tmp/new.so pixie_amd/libpixie_hip.so
for sc in jelly sand jelly100k; do
  if [ $sc = jelly ]; then CMD="env PIXIE_MPM_WARM=100 python $ROOT/scripts/mpm_bench.py 1000000 120 300"; elif [ $sc = sand ]; then CMD="env PIXIE_MPM_SCENARIO=sand PIXIE_MPM_WARM=100 python $ROOT/scripts/mpm_bench.py 1000000 0 400"; else CMD="env PIXIE_MPM_WARM=100 python $ROOT/scripts/mpm_bench.py 100000 50 1800"; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_$sc -o mpm -- $CMD > $ROOT/$OUT/run_$sc.txt 2>&1)
  DB=$(find $OUT/prof_$sc -name "*.db" | head -1)
  [ -n "$DB" ] && python scripts/rocpd_stats.py $DB $OUT/stats_$sc.csv
  rm -rf $OUT/prof_$sc
done
timeout 1500 python -m pytest tests/test_mpm_hip.py tests/test_mpm_ref_hip.py tests/test_pipeline_hip.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
python - <<'PY'
import re, csv
for l in open('gpurun_out/r6q/ab.txt'):
    m=re.search(r'^(\S+) n=(\d+) ng=(\d+) .* (tree|sand|metal) .*: ([\d.]+) us/substep', l)
    if m: print(m.groups())
for sc in ('jelly','sand','jelly100k'):
    for r in csv.DictReader(open(f'gpurun_out/r6q/stats_{sc}.csv')):
        if 'mpm_grid_block' in r['kernel'] or 'mpm_block_kernel<true, true' in r['kernel']: print(sc, r['kernel'][12:50], r['calls'], r['avg_us'])
PY
cat $OUT/state_hash.txt; tail -3 $OUT/pytest.log
