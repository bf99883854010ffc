#!/bin/bash
# (Record of an experiment.  The comparison libraries under scripts/_ab/ are not kept: check out the commit named in the matching profiles/ file,
# run `python -m pixie_amd.build`, and copy pixie_amd/libpixie_hip.so there under the name this script expects.)
# Did the polar-iteration variant (commit 9da2b5a, scripts/_ab/libpixie_hip_polarB.so) execute fewer VALU instructions than the shipped
# kernel?  SQ_INSTS_VALU / SQ_WAVES / SQ_ACTIVE_INST_VALU of the 1 M step loop under both libraries (counters only: no trace domains).
OUT=gpurun_out/${1:-r4w}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$(pwd)
cp pixie_amd/libpixie_hip.so /tmp/ship.so
run_pmc () {  # name, counters..., -- cmd
  local name=$1; shift
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  (cd /tmp && timeout 300 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$name -o $name -- "$@" > $ROOT/$OUT/pmc_$name.log 2>&1)
  local f=$(find $OUT/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python scripts/pmc_summary.py $f > $OUT/pmc_$name.txt; else echo "no counter csv for $name" > $OUT/pmc_$name.txt; tail -5 $OUT/pmc_$name.log >> $OUT/pmc_$name.txt; fi
  rm -rf $OUT/pmc_$name $OUT/pmc_$name.log
}
M1M="python $ROOT/scripts/mpm_bench.py 1000000 120 60"
for which in ship polarB; do
  if [ $which = ship ]; then cp /tmp/ship.so pixie_amd/libpixie_hip.so; else cp scripts/_ab/libpixie_hip_polarB.so pixie_amd/libpixie_hip.so; fi
  PIXIE_MPM_WARM=100 run_pmc ${which}_sq SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS -- $M1M
  PIXIE_MPM_V0=0.6 PIXIE_MPM_WARM=300 run_pmc ${which}_sq_in_motion SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS -- $M1M
done
cp /tmp/ship.so pixie_amd/libpixie_hip.so
for f in $OUT/pmc_*.txt; do echo "== $f"; grep mpm_block_kernel $f | cut -c1-400; done
