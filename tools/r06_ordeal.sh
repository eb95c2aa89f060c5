#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_ordeal
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_ordeal.py tests/test_storytelling.py tests/test_generic_specialised.py tests/test_random_directive_games.py tests/test_random_prefab_games.py tests/test_checkpoint.py tests/test_reference_known_answers.py -m gpu -x -q 2>&1 | tail -40 > $OUT/tests.txt
cat $OUT/tests.txt
