#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_jit_shapes2
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_scrolly_specialised.py tests/test_generic_persistent.py -m gpu -q -x 2>&1 | tail -4 > $OUT/tests.txt
PCX_SM_JIT=1 python -m pytest tests/test_hip_parity.py tests/test_cropping.py tests/test_postprocess.py tests/test_random_levels.py tests/test_checkpoint.py tests/test_reference_known_answers.py -m gpu -q -k "scrolly or Scrolly or scroll" 2>&1 | tail -6 >> $OUT/tests.txt
cat $OUT/tests.txt
