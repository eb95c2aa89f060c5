#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_persistent_shapes.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r04/call2_tests.txt
cat gpurun_out/r04/call2_tests.txt
timeout 1200 python tools/ps_sweep.py --prof --batches 131072,1048576 --variants head,A,A_static,A_cu6,A_cu5,B,B_cu2,B_u32_cu5 --extra A_cu5:PCX_SM_SHAPE=1+PCX_SM_PER_CU=5 --steps 100 --repeats 3 --out gpurun_out/r04/ps_sweep2.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/ps_sweep2.txt | tail -70
