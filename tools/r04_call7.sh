#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 1200 python tools/ps_sweep.py --prof --batches 131072,1048576 --variants head,C,C_prio,Cs,A_static --steps 80 --repeats 3 --out gpurun_out/r04/ps_sweep7.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/ps_sweep7.txt | tail -80
