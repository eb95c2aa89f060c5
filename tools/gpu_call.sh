#!/bin/bash
# scratch driver for one gpurun call (overwritten per experiment)
OUT=gpurun_out/r03_call30; mkdir -p $OUT
for w in default 0 1; do
  echo "== PCX_GENERIC_LATE_WB=$w"
  if [ $w = default ]; then timeout 600 python tools/generic_timing.py 2>&1 | grep pcx_generic
  else PCX_GENERIC_LATE_WB=$w timeout 600 python tools/generic_timing.py 2>&1 | grep pcx_generic; fi
done > $OUT/generic_timing.txt 2>&1
cat $OUT/generic_timing.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/suite.log 2>&1; echo "rc=$?" >> $OUT/suite.log; tail -4 $OUT/suite.log
