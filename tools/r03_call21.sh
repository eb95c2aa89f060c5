#!/bin/bash
OUT=gpurun_out/r03_call21; mkdir -p $OUT
timeout 900 python -m pytest tests/test_cropping.py tests/test_reference_known_answers.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log
tail -5 $OUT/tests.log
for c in warehouse_L0:262144 walkers_scroll_groups:262144 hello_world:262144 better_scrolly_custom_B:262144 marauders:262144 directives_z_order:262144; do
  for pw in 0 4 6 8 10; do echo -n "PIPE=$pw "; PCX_GENERIC_PIPE=$pw timeout 120 python tools/generic_timing.py $c 2>&1 | grep pcx_generic; done
  for pw in 4 8; do echo -n "PIPE=$pw WAVES=3 "; PCX_GENERIC_PIPE_WAVES=3 PCX_GENERIC_PIPE=$pw timeout 120 python tools/generic_timing.py $c 2>&1 | grep pcx_generic; done
done > $OUT/pipe.txt 2>&1
cat $OUT/pipe.txt
PCX_GENERIC_PIPE=8 timeout 900 python -m pytest tests/test_hip_parity.py tests/test_random_levels.py tests/test_reference_known_answers.py -m gpu -x -q -k "generic or table or random or known or walkers or directive" > $OUT/pipe_tests.log 2>&1; echo "rc=$?" >> $OUT/pipe_tests.log
tail -5 $OUT/pipe_tests.log
