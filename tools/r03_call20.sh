#!/bin/bash
OUT=gpurun_out/r03_call20; mkdir -p $OUT
timeout 900 python -m pytest tests/test_cropping.py tests/test_reference_known_answers.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log
tail -30 $OUT/tests.log
