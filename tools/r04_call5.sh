#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
export PCX_SM_TRACE=1
timeout 1200 python tools/ps_sweep.py --prof --batches 131072,1048576 --variants head,A_static,Cs,Cs_nolock,Cs_w3,Cs_w4,Cs_w6,Cs_w1_cu6,C --steps 60 --repeats 3 --out gpurun_out/r04/ps_sweep5.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/ps_sweep5.txt | tail -80
