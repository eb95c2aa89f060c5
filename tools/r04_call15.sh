#!/bin/bash
# generic kernel: baseline of round 4 (ten fixtures), logic-only ablation and phase timers on the three VERDICT fixtures
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_generic; mkdir -p $OUT
cd $ROOT
python tools/generic_timing.py > $OUT/baseline.txt 2>&1
for fx in warehouse_L0:262144 marauders_custom_A:32768 walkers_scroll_groups:262144 directives_z_order:262144; do
  echo "== $fx logic only (PCX_DEBUG=2)" >> $OUT/ablate.txt
  PCX_DEBUG=2 python tools/generic_timing.py $fx >> $OUT/ablate.txt 2>&1
  echo "== $fx phase timers (PCX_DEBUG=8)" >> $OUT/ablate.txt
  PCX_DEBUG=8 python tools/generic_timing.py $fx 2>&1 | tail -4 >> $OUT/ablate.txt
  echo "== $fx logic only + phase timers (PCX_DEBUG=10)" >> $OUT/ablate.txt
  PCX_DEBUG=10 python tools/generic_timing.py $fx 2>&1 | tail -4 >> $OUT/ablate.txt
done
cat $OUT/baseline.txt $OUT/ablate.txt
