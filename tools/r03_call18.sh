#!/bin/bash
OUT=gpurun_out/r03_call18; mkdir -p $OUT
for tp in 0 1 0 1; do PCX_EPI_TWO_PASS=$tp timeout 300 python tools/fusion_bench.py sm >> $OUT/sm_epi.txt 2>&1; done
timeout 600 python -m pytest tests/test_postprocess.py -m gpu -x -q > $OUT/post_tests.log 2>&1; echo "rc=$?" >> $OUT/post_tests.log
tail -3 $OUT/post_tests.log; cat $OUT/sm_epi.txt
