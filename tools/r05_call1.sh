#!/bin/bash
# round 5, call 1: the baked-constants instance of pcx_scrolly_maze_step -- parity, then same-box A/B with phase timers, then the bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call1; mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_persistent_shapes.py -x -q -m gpu -k "compiled_in or other_levels or config_5 or semaphore" > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt
tail -5 $OUT/pytest.txt
timeout 900 python tools/ps_sweep.py --batches 131072,262144,1048576 --steps 100 --repeats 3 --prof \
  --variants auto,unbaked,head \
  --extra "cu4:PCX_SM_PER_CU=4,cu5:PCX_SM_PER_CU=5,cu2:PCX_SM_PER_CU=2,w3x2:PCX_SM_WAVES=3+PCX_SM_PER_CU=2,w3x3:PCX_SM_WAVES=3+PCX_SM_PER_CU=3,lock2:PCX_SM_LOCK=2,w4x2k2:PCX_SM_WAVES=4+PCX_SM_PER_CU=2+PCX_SM_LOCK=2,dyn:PCX_SM_DYNAMIC=1,static:PCX_SM_DYNAMIC=0,d1:PCX_DEBUG=1,d4:PCX_DEBUG=4,d5:PCX_DEBUG=5" \
  --out $OUT/ps_sweep.json > $OUT/ps_sweep.txt 2>&1
cat $OUT/ps_sweep.txt | tail -80
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json
