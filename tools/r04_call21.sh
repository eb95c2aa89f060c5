#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_spec; mkdir -p $OUT
cd $ROOT
timeout 2700 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/suite.txt 2>&1; echo "suite rc=$?"; tail -6 $OUT/suite.txt | cut -c1-300
