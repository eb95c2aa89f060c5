#!/bin/bash
# scratch driver for one gpurun call (overwritten per experiment)
OUT=gpurun_out/r03_call37; mkdir -p $OUT
for tp in 0 1 0 1; do echo "== PCX_EPI_TWO_PASS=$tp"; PCX_EPI_TWO_PASS=$tp timeout 600 python tools/fusion_bench.py hwc 2>&1 | grep "channels last"; done > $OUT/hwc.txt 2>&1; cat $OUT/hwc.txt
