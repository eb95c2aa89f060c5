#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_cropping.py tests/test_postprocess.py tests/test_checkpoint.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r04/call14_tests.txt
cat gpurun_out/r04/call14_tests.txt
timeout 600 python tools/crop_features_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/crop_features_bench.txt
