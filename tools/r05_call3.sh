#!/bin/bash
# round 5, call 3: board-dword pairs (global_store_dwordx2) in the persistent owner-code shapes -- parity, then the shape sweep again
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call3; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_persistent_shapes.py -x -q -m gpu > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt
tail -5 $OUT/pytest.txt
timeout 900 python tools/ps_sweep.py --batches 131072,262144,1048576 --steps 100 --repeats 3 --prof \
  --variants auto,unbaked \
  --extra "nopairs:PCX_SM_PAIRS=0,cu2:PCX_SM_PER_CU=2,cu2np:PCX_SM_PER_CU=2+PCX_SM_PAIRS=0,cu3:PCX_SM_PER_CU=3,cu4:PCX_SM_PER_CU=4,w3x1k1:PCX_SM_WAVES=3+PCX_SM_PER_CU=1,w4x1k1:PCX_SM_WAVES=4+PCX_SM_PER_CU=1,w5x1k1:PCX_SM_WAVES=5+PCX_SM_PER_CU=1,w6x1k1:PCX_SM_WAVES=6+PCX_SM_PER_CU=1,w4x1k2:PCX_SM_WAVES=4+PCX_SM_PER_CU=1+PCX_SM_LOCK=2,w6x1k2:PCX_SM_WAVES=6+PCX_SM_PER_CU=1+PCX_SM_LOCK=2,w8x1k2:PCX_SM_WAVES=8+PCX_SM_PER_CU=1+PCX_SM_LOCK=2,w3x2:PCX_SM_WAVES=3+PCX_SM_PER_CU=2,w3x3:PCX_SM_WAVES=3+PCX_SM_PER_CU=3,w4x2k2:PCX_SM_WAVES=4+PCX_SM_PER_CU=2+PCX_SM_LOCK=2,w2x2k0:PCX_SM_PER_CU=2+PCX_SM_LOCK=0,w1x4:PCX_SM_WAVES=1+PCX_SM_PER_CU=4+PCX_SM_LOCK=0,d5:PCX_DEBUG=5" \
  --out $OUT/ps_sweep.json > $OUT/ps_sweep.txt 2>&1
grep -v "^ *\[" $OUT/ps_sweep.txt | tail -70
