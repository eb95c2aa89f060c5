#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "several_steps or step_n" 2>&1 | tail -6 > gpurun_out/r04/call13_tests.txt
cat gpurun_out/r04/call13_tests.txt
timeout 600 python tools/step_n_timing.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/step_n_timing.txt
