#!/bin/bash
# round 5, call 5: tickets that steal across the work counter's shards -- parity, then A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call5; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_persistent_shapes.py -x -q -m gpu -k "compiled_in or semaphore or shapes_match or small_units or headline" > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt
tail -4 $OUT/pytest.txt
timeout 900 python tools/ps_sweep.py --batches 131072,262144,524288,1048576 --steps 100 --repeats 3 --prof \
  --variants auto,unbaked \
  --extra "nosteal:PCX_SM_STEAL=0,dyn:PCX_SM_DYNAMIC=1,dynt1:PCX_SM_DYNAMIC=1+PCX_SM_TAIL=1,dynt1u32:PCX_SM_DYNAMIC=1+PCX_SM_TAIL=1+PCX_SM_TAIL_UNIT=32,cu2:PCX_SM_PER_CU=2,cu2d:PCX_SM_PER_CU=2+PCX_SM_DYNAMIC=1,w4x1k2:PCX_SM_WAVES=4+PCX_SM_PER_CU=1+PCX_SM_LOCK=2,w4x1k2d:PCX_SM_WAVES=4+PCX_SM_PER_CU=1+PCX_SM_LOCK=2+PCX_SM_DYNAMIC=1,w6x1k3:PCX_SM_WAVES=6+PCX_SM_PER_CU=1+PCX_SM_LOCK=3,w6x1k3d:PCX_SM_WAVES=6+PCX_SM_PER_CU=1+PCX_SM_LOCK=3+PCX_SM_DYNAMIC=1,w3x3:PCX_SM_WAVES=3+PCX_SM_PER_CU=3,w3x3d:PCX_SM_WAVES=3+PCX_SM_PER_CU=3+PCX_SM_DYNAMIC=1,w4x2k2:PCX_SM_WAVES=4+PCX_SM_PER_CU=2+PCX_SM_LOCK=2,w4x2k2d:PCX_SM_WAVES=4+PCX_SM_PER_CU=2+PCX_SM_LOCK=2+PCX_SM_DYNAMIC=1,w8x1k3:PCX_SM_WAVES=8+PCX_SM_PER_CU=1+PCX_SM_LOCK=3,w8x1k3d:PCX_SM_WAVES=8+PCX_SM_PER_CU=1+PCX_SM_LOCK=3+PCX_SM_DYNAMIC=1,w8x1k4d:PCX_SM_WAVES=8+PCX_SM_PER_CU=1+PCX_SM_LOCK=4+PCX_SM_DYNAMIC=1" \
  --out $OUT/ps_sweep.json > $OUT/ps_sweep.txt 2>&1
grep -v "^ *\[" $OUT/ps_sweep.txt | tail -84
timeout 300 python bench.py --no-other-configs --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json
for l in open('$OUT/bench.json'):
  if l.startswith('{'):
    d=json.loads(l); print(d['ms_per_step'], d['roofline'])"
