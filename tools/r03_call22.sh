#!/bin/bash
OUT=gpurun_out/r03_call22; mkdir -p $OUT
timeout 900 python -m pytest tests/test_cropping.py tests/test_reference_known_answers.py -m gpu -q > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log
tail -5 $OUT/tests.log
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -k "table_driven" > $OUT/tests2.log 2>&1; echo "rc=$?" >> $OUT/tests2.log
tail -5 $OUT/tests2.log
timeout 600 python tools/generic_timing.py > $OUT/generic_timing.txt 2>&1
cat $OUT/generic_timing.txt
