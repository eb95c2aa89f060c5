#!/bin/bash
OUT=gpurun_out/r03_call24; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/suite.log 2>&1; echo "suite rc=$?" >> $OUT/suite.log
tail -6 $OUT/suite.log
timeout 600 python tools/generic_timing.py > $OUT/generic_timing.txt 2>&1
cat $OUT/generic_timing.txt
