#!/bin/bash
# the whole GPU suite + smoke, as the driver runs them at round end
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_suite; mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -q -m gpu --durations=25 > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt
tail -40 $OUT/pytest.txt
python __graft_entry__.py smoke 2>&1 | tail -2
