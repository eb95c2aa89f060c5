#!/bin/bash
# round 5, call 4: launch-shape sweep of the baked instance (dword loop): workers per CU, slots, tickets, small tail units
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call4; mkdir -p $OUT
cd $ROOT
timeout 900 python tools/ps_sweep.py --batches 131072,262144,524288,1048576 --steps 100 --repeats 3 --prof \
  --variants auto,unbaked \
  --extra "cu2:PCX_SM_PER_CU=2,cu2s:PCX_SM_PER_CU=2+PCX_SM_DYNAMIC=0,cu2d:PCX_SM_PER_CU=2+PCX_SM_DYNAMIC=1,cu2t1:PCX_SM_PER_CU=2+PCX_SM_DYNAMIC=1+PCX_SM_TAIL=1,cu2t2:PCX_SM_PER_CU=2+PCX_SM_DYNAMIC=1+PCX_SM_TAIL=2,cu2t1u32:PCX_SM_PER_CU=2+PCX_SM_DYNAMIC=1+PCX_SM_TAIL=1+PCX_SM_TAIL_UNIT=32,w4x1k2:PCX_SM_WAVES=4+PCX_SM_PER_CU=1+PCX_SM_LOCK=2,w4x1k1:PCX_SM_WAVES=4+PCX_SM_PER_CU=1,w3x1k1:PCX_SM_WAVES=3+PCX_SM_PER_CU=1,w5x1k2:PCX_SM_WAVES=5+PCX_SM_PER_CU=1+PCX_SM_LOCK=2,w6x1k2:PCX_SM_WAVES=6+PCX_SM_PER_CU=1+PCX_SM_LOCK=2,w6x1k3:PCX_SM_WAVES=6+PCX_SM_PER_CU=1+PCX_SM_LOCK=3,w3x2:PCX_SM_WAVES=3+PCX_SM_PER_CU=2,w3x3:PCX_SM_WAVES=3+PCX_SM_PER_CU=3,w4x2k2:PCX_SM_WAVES=4+PCX_SM_PER_CU=2+PCX_SM_LOCK=2,w8x1k2:PCX_SM_WAVES=8+PCX_SM_PER_CU=1+PCX_SM_LOCK=2,w8x1k3:PCX_SM_WAVES=8+PCX_SM_PER_CU=1+PCX_SM_LOCK=3,cu2prio:PCX_SM_PER_CU=2+PCX_SM_PRIO=1" \
  --out $OUT/ps_sweep.json > $OUT/ps_sweep.txt 2>&1
grep -v "^ *\[" $OUT/ps_sweep.txt | tail -90
