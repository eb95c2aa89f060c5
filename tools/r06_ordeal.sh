#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_ordeal
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_ordeal.py -m gpu -x -q 2>&1 | tail -40 > $OUT/tests.txt
cat $OUT/tests.txt
