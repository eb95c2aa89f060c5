#!/bin/bash
# pcx_generic_step (specialised build, the default at these sizes) under rocprofv3 at the three fixtures VERDICT r3 names:
# per fixture one --kernel-trace --stats run and four separate --pmc passes; plus the phase timers of both builds.
#   gpurun -- bash tools/profile_generic_r04.sh   ->  gpurun_out/prof_generic_r04/{summary.txt,<fixture>/...}
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_generic_r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/summary.txt
python $ROOT/tools/generic_timing.py warehouse_L0:4096 marauders_custom_A:4096 walkers_scroll_groups:4096 > /dev/null 2>&1   # (fills the code-object cache)
for fx in warehouse_L0:262144 marauders_custom_A:32768 walkers_scroll_groups:262144; do
  name=${fx%%:*}
  D=$OUT/$name; mkdir -p $D
  CMD="python $ROOT/tools/generic_timing.py $fx"
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o t -- $CMD > $D/trace.log 2>&1
  for pass in "write WRITE_SIZE" "fetch FETCH_SIZE" "sqa SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "sqb SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY"; do
    set -- $pass; p=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $D/pmc_$p -o p -- $CMD > $D/pmc_$p.log 2>&1
  done
  echo "== $fx (specialised build)" >> $OUT/summary.txt
  grep pcx_generic $D/trace.log | tail -1 >> $OUT/summary.txt
  f=$(find $D/trace -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && { head -1 $f; grep generic $f; } | cut -c1-260 >> $OUT/summary.txt
  python $ROOT/tools/pmc_summary.py $D pcx_generic >> $OUT/summary.txt 2>&1
  echo "-- phase timers (PCX_DEBUG=8), specialised / table-driven" >> $OUT/summary.txt
  PCX_DEBUG=8 python $ROOT/tools/generic_timing.py $fx 2>&1 | grep "cycles per group" | tail -1 >> $OUT/summary.txt
  PCX_GENERIC_JIT=0 PCX_DEBUG=8 python $ROOT/tools/generic_timing.py $fx 2>&1 | grep "cycles per group" | tail -1 >> $OUT/summary.txt
done
echo "== directives_z_order:262144 kernel time, specialised" >> $OUT/summary.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dz -o t -- python $ROOT/tools/generic_timing.py directives_z_order:262144 > $OUT/dz.log 2>&1
grep pcx_generic $OUT/dz.log | tail -1 >> $OUT/summary.txt
f=$(find $OUT/dz -name '*kernel_stats.csv' | head -1); grep generic $f | cut -c1-200 >> $OUT/summary.txt
cat $OUT/summary.txt
