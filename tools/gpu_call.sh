#!/bin/bash
# scratch driver for one gpurun call (overwritten per experiment)
OUT=gpurun_out/r03_call31; mkdir -p $OUT
timeout 900 python -m pytest tests/test_postprocess.py -m gpu -q -x > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log; tail -15 $OUT/tests.log
timeout 900 python -m pytest tests/test_random_levels.py -m gpu -q -x -k "fused_croppers" > $OUT/tests2.log 2>&1; echo "rc=$?" >> $OUT/tests2.log; tail -8 $OUT/tests2.log
