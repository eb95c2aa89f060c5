#!/bin/bash
# round 5, call 14: launches of several steps with persistent workers (shape 13): parity, then timing
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call14; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_persistent_shapes.py -q -m gpu -x -k "several_steps_per_launch or compiled_in or semaphore or headline or config_5 or other_levels or shapes_match" > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt
tail -4 $OUT/pytest.txt
echo done
