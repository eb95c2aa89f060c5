#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests/test_checkpoint.py tests/test_distributed.py tests/test_postprocess.py tests/test_storytelling.py tests/test_cropping.py -x -q -m gpu 2>&1 | tail -25 ) > gpurun_out/r04/call11_tests.txt 2>&1
cat gpurun_out/r04/call11_tests.txt
