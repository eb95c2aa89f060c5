#!/bin/bash
OUT=gpurun_out/r03_call26; mkdir -p $OUT
for c in warehouse_L0:262144 hello_world:262144 marauders:32768 walkers_scroll_groups:262144 better_scrolly_custom_B:262144 marauders_custom_A:32768; do
  for v in 5 6 8; do
    for w in default 2 3 4; do
      echo -n "gen$v WAVES=$w "
      if [ $w = default ]; then PCX_LIB=$PWD/gpurun_variants/libpcx_gen$v.so timeout 120 python tools/generic_timing.py $c 2>&1 | grep pcx_generic
      else PCX_GENERIC_WAVES=$w PCX_LIB=$PWD/gpurun_variants/libpcx_gen$v.so timeout 120 python tools/generic_timing.py $c 2>&1 | grep pcx_generic; fi
    done
  done
done > $OUT/occ_ab.txt 2>&1
cat $OUT/occ_ab.txt
