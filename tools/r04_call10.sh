#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_persistent_shapes.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r04/call10_tests.txt
cat gpurun_out/r04/call10_tests.txt
timeout 900 python tools/ps_sweep.py --batches 262144,524288,1048576 --variants head,auto,auto_t0,auto_t1,auto_t3,auto_t2_u32,auto_t2_u8,auto_dyn,auto_dyn_t0,auto_static --steps 60 --repeats 3 --out gpurun_out/r04/ps_sweep10.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/ps_sweep10.txt | tail -32
