#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_call8
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_cropping.py tests/test_random_levels.py tests/test_reference_known_answers.py -m gpu -q > $OUT/crop_tests.log 2>&1; echo "crop tests rc=$?"; grep -E "passed|failed|FAILED" $OUT/crop_tests.log | tail -6
for lib in head new; do
  if [ $lib = head ]; then export PCX_LIB=$ROOT/gpurun_variants/libpcx_head.so; else unset PCX_LIB; fi
  echo "== library $lib"; timeout 600 python tools/post_bench.py 2>/dev/null | grep -E "Cropper|croppers|windows only" 
done > $OUT/post_ab.txt 2>&1
cat $OUT/post_ab.txt
