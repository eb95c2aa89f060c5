#!/bin/bash
# round 4, GPU call 1: parity of the persistent shapes, then the same-box sweep
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_persistent_shapes.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04/call1_tests.txt
cat gpurun_out/r04/call1_tests.txt
timeout 1200 python tools/ps_sweep.py --batches 131072,262144,1048576 --steps 100 --repeats 3 --out gpurun_out/r04/ps_sweep1.json 2>&1 | tee gpurun_out/r04/ps_sweep1.txt | tail -70
