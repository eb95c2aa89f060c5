#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 1200 python tools/ps_sweep.py --prof --batches 1048576 --variants head,A,A_static --extra "A_d1:PCX_SM_SHAPE=1+PCX_DEBUG=1,A_d4:PCX_SM_SHAPE=1+PCX_DEBUG=4,A_d5:PCX_SM_SHAPE=1+PCX_DEBUG=5,As_d5:PCX_SM_SHAPE=1+PCX_DEBUG=5+PCX_SM_DYNAMIC=0,head_d5:PCX_DEBUG=5,A_cu5:PCX_SM_SHAPE=1+PCX_SM_PER_CU=5,A_cu4:PCX_SM_SHAPE=1+PCX_SM_PER_CU=4,A_cu3:PCX_SM_SHAPE=1+PCX_SM_PER_CU=3,A_cu4_d5:PCX_SM_SHAPE=1+PCX_SM_PER_CU=4+PCX_DEBUG=5,A_cu2_d5:PCX_SM_SHAPE=1+PCX_SM_PER_CU=2+PCX_DEBUG=5" --steps 60 --repeats 3 --out gpurun_out/r04/ps_sweep3.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/ps_sweep3.txt | tail -70
