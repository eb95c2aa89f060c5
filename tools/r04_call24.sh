#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_spec; mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_postprocess.py -q -m gpu -p no:cacheprovider > $OUT/post.txt 2>&1; echo "rc=$?"; tail -5 $OUT/post.txt | cut -c1-250
grep -n "^E  " $OUT/post.txt | cut -c1-200 | head -20
