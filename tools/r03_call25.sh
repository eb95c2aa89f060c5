#!/bin/bash
OUT=gpurun_out/r03_call25; mkdir -p $OUT
for c in warehouse_L0:262144 warehouse_L0:1048576 hello_world:262144 marauders:32768 walkers_scroll_groups:262144 better_scrolly_custom_B:262144; do
  for d in 0 16 2 18; do echo -n "PCX_DEBUG=$d "; PCX_DEBUG=$d timeout 120 python tools/generic_timing.py $c 2>&1 | grep pcx_generic; done
done > $OUT/store_ablation.txt 2>&1
cat $OUT/store_ablation.txt
