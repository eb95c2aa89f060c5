#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_persistent_shapes.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r04/call4_tests.txt
cat gpurun_out/r04/call4_tests.txt
timeout 1200 python tools/ps_sweep.py --prof --batches 131072,262144,1048576 --variants head,A,A_static,C,C_nolock,C_static,C_cu2,C_u32_w4,C_u32_w3_cu3,C_u32_w2_cu4 --steps 60 --repeats 3 --out gpurun_out/r04/ps_sweep4.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/ps_sweep4.txt | tail -80
