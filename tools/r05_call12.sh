#!/bin/bash
# round 5, call 12: pcx_hello_world_step with persistent workers -- parity, then the sweep
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call12; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_persistent_shapes.py -q -m gpu -x -k "stream_kernels" > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt
tail -4 $OUT/pytest.txt
timeout 600 python tools/wm_sweep.py --game hello_world --batches 262144,1048576 --variants pw0,auto,w5k4,w5k3,w5k2,w5k0,w4k2,w3k2x2,w2k1x3,w2k1x4,w2k0x3,w1k0x4,w1k0x6,w1k0x8,w3k0x2 > $OUT/hw_sweep.txt 2>&1; grep -v amdgpu.ids $OUT/hw_sweep.txt | tail -32
