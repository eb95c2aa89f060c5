#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_persistent_shapes.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r04/call8_tests.txt
cat gpurun_out/r04/call8_tests.txt
timeout 1200 python tools/ps_sweep.py --prof --batches 131072,262144,1048576 --variants head,C2x3,C2x3s,C2x4,C2x4s,C3x3,C3x3s,C3x3_k2,C4x2_k2,C4x2_k2s,A8s,A --steps 60 --repeats 3 --out gpurun_out/r04/ps_sweep8.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/ps_sweep8.txt | tail -80
