#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_call10
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_cropping.py tests/test_random_levels.py tests/test_reference_known_answers.py tests/test_checkpoint.py tests/test_distributed.py -m gpu -q -x > $OUT/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FAILED" $OUT/tests.log | tail -6
bash tools/small_batch_ablation.sh 2>&1 | head -7
