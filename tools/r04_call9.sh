#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r04/call9_tests.txt 2>&1
cat gpurun_out/r04/call9_tests.txt
timeout 900 python tools/ps_sweep.py --batches 65536,131072,262144,524288,1048576,2097152 --variants auto,head --steps 60 --repeats 3 --out gpurun_out/r04/ps_sweep9.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/ps_sweep9.txt | tail -20
