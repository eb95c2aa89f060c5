#!/bin/bash
# round 5, call 10: warehouse persistent workers: the remaining parity cases, more slot configurations
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call10; mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_persistent_shapes.py -q -m gpu -x -k "warehouse" > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt
tail -4 $OUT/pytest.txt
timeout 600 python tools/wm_sweep.py --game warehouse --batches 262144,1048576 --variants pw0,w6k3,w2k1x3,w8k4,w6k4,w8k0,w6k0,w4k0x2,w4k3x2,w3k2x3,w3k0x3,w2k0x3,w2k2x4,w2k1x4,w1k0x4,w1k0x6,w8k3d,w8k4s > $OUT/wm_sweep.txt 2>&1; grep -v amdgpu.ids $OUT/wm_sweep.txt | tail -40
