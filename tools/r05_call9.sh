#!/bin/bash
# round 5, call 9: pcx_warehouse_step with persistent workers -- parity, then the sweep
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call9; mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_persistent_shapes.py -q -m gpu -x -k "warehouse" > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt
tail -6 $OUT/pytest.txt
timeout 600 python tools/wm_sweep.py --game warehouse --batches 262144,1048576 > $OUT/wm_sweep.txt 2>&1; grep -v amdgpu.ids $OUT/wm_sweep.txt | tail -32
