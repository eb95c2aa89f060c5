#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_persistent_shapes.py -x -q -m gpu -k "semaphore" 2>&1 | tail -8 > gpurun_out/r04/call6_tests.txt
cat gpurun_out/r04/call6_tests.txt
timeout 1200 python tools/ps_sweep.py --prof --batches 131072,1048576 --variants head,head_mask,C,C_w6x1_k3,C_w6x1_k2,M_w4x3_k1,M_w4x3_k2,M_w3x3_k1,M_w3x4_k1,M_w6x2_k2,M_w6x2_k1,M_w12x1_k3,M_w12x1_k4,M_w12x1_k0,M_w2x4_k1,M1_cu8,M1_cu12 --steps 60 --repeats 3 --out gpurun_out/r04/ps_sweep6.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/ps_sweep6.txt | tail -80
