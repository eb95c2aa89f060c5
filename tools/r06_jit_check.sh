#!/bin/bash
# Round 6: the run-time instances of pcx_scrolly_maze_step (a level of one's own on the example's board): parity, then same-box timing
# against the run-time-constants instances and against the shipped level 0 (whose instances are in libpcx.so).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_jit
mkdir -p $OUT
cd $ROOT
Q="grep -v amdgpu.ids"
python -m pytest tests/test_scrolly_specialised.py tests/test_persistent_shapes.py -m gpu -q -x 2>&1 | tail -5 > $OUT/tests.txt
python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "custom_H" 2>&1 | tail -3 >> $OUT/tests.txt
{
echo "# scrolly_custom_H (10x30 board, abcP, own maze): run-time build (launch shape 7) against the run-time-constants instance (3)"
python tools/env_sweep.py --fixture scrolly_custom_H --batches 131072,1048576 --steps 40 --variants "jit;nojit:!PCX_SM_JIT=0" 2>&1 | $Q
echo "# the shipped level 0 (launch shape 5: the instance in libpcx.so), same box"
python tools/env_sweep.py --game scrolly_maze --batches 131072,1048576 --steps 40 --variants "auto" 2>&1 | $Q
echo "# small batch, the cooperative instance: 4,096 environments"
python tools/env_sweep.py --fixture scrolly_custom_H --batches 4096 --steps 200 --variants "jit;nojit:!PCX_SM_JIT=0" 2>&1 | $Q
python tools/env_sweep.py --game scrolly_maze --batches 4096 --steps 200 --variants "auto" 2>&1 | $Q
} > $OUT/r06_scrolly_jit_timing.txt 2>&1
cat $OUT/tests.txt; tail -30 $OUT/r06_scrolly_jit_timing.txt
