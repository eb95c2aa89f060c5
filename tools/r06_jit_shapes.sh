#!/bin/bash
# Round 6: pcx_scrolly_maze_step's run-time instances for levels of OTHER board shapes and casts (launch shape 21): parity, then
# same-box timing against the shape-generic instances of libpcx.so (launch shape 20).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_jit_shapes
mkdir -p $OUT
cd $ROOT
Q="grep -v amdgpu.ids"
python -m pytest tests/test_scrolly_specialised.py -m gpu -q -x 2>&1 | tail -4 > $OUT/tests.txt
PCX_SM_JIT=1 python -m pytest tests/test_hip_parity.py tests/test_cropping.py tests/test_postprocess.py tests/test_random_levels.py -m gpu -q -x -k "scrolly" 2>&1 | tail -3 >> $OUT/tests.txt
cat $OUT/tests.txt
{
for f in scrolly_custom_A scrolly_custom_B scrolly_custom_E scrolly_custom_F scrolly_custom_G scrolly_custom_A_unoccluded; do
  python tools/env_sweep.py --fixture $f --batches 131072 --steps 40 --variants "jit;generic:!PCX_SM_JIT=0" 2>&1 | $Q
done
python tools/env_sweep.py --fixture scrolly_custom_G --batches 1048576 --steps 20 --variants "jit;generic:!PCX_SM_JIT=0" 2>&1 | $Q
} > $OUT/r06_scrolly_jit_other_shapes.txt 2>&1
cat $OUT/r06_scrolly_jit_other_shapes.txt
