#!/bin/bash
OUT=gpurun_out/r03_call23; mkdir -p $OUT
timeout 900 python -m pytest tests/test_cropping.py tests/test_reference_known_answers.py -m gpu -q > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log
tail -5 $OUT/tests.log
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -k "table_driven" > $OUT/tests2.log 2>&1; echo "rc=$?" >> $OUT/tests2.log
tail -5 $OUT/tests2.log
for c in warehouse_L0:262144 hello_world:262144 better_scrolly_custom_B:262144 walkers_scroll_groups:262144 marauders:262144 warehouse_L0:1048576; do
  for tk in 0 2 3 4 5 6; do echo -n "TOKENS=$tk "; PCX_GENERIC_TOKENS=$tk timeout 120 python tools/generic_timing.py $c 2>&1 | grep pcx_generic; done
done > $OUT/tokens.txt 2>&1
cat $OUT/tokens.txt
