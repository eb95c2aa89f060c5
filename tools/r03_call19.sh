#!/bin/bash
OUT=gpurun_out/r03_call19; mkdir -p $OUT
for c in warehouse_L0:262144 walkers_scroll_groups:262144 marauders:32768 marauders_custom_A:32768; do
  echo "== $c"
  PCX_DEBUG=8 timeout 120 python tools/generic_timing.py $c 2>&1 | grep -E "cycles per group|pcx_generic" | tail -2
  for d in 0 1 2 3; do echo -n "PCX_DEBUG=$d "; PCX_DEBUG=$d timeout 120 python tools/generic_timing.py $c 2>&1 | grep pcx_generic; done
  for w in 1 2 4 8; do echo -n "WAVES=$w "; PCX_GENERIC_WAVES=$w timeout 120 python tools/generic_timing.py $c 2>&1 | grep pcx_generic; done
done > $OUT/generic_cycles.txt 2>&1
cat $OUT/generic_cycles.txt
