#!/bin/bash
# round 5, call 7: the pruned kernel file (two launch shapes gone, no pair variants), the launch-shape tuner
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call7; mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_persistent_shapes.py tests/test_hip_parity.py tests/test_raise_parity.py tests/test_distributed.py -q -m gpu -x > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt
tail -5 $OUT/pytest.txt
for B in 131072 262144 1048576; do
  PCX_DEBUG=16 timeout 300 python bench.py --batch $B --steps 100 --warmup 30 --repeats 3 --no-other-configs --no-cpu-baseline 2> $OUT/tune_$B.err | python -c "
import json,sys
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); print($B, 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'achievable', d['roofline'].get('achievable',{}).get('GBps'))"
  grep "pcx scrolly" $OUT/tune_$B.err | tail -2
done
timeout 900 python tools/ps_sweep.py --batches 65536,131072,262144,1048576 --steps 100 --repeats 3 \
  --variants auto,unbaked \
  --extra "s3:PCX_SM_SHAPE=3,s0:PCX_SM_SHAPE=0,w4x1k2:PCX_SM_WAVES=4+PCX_SM_PER_CU=1+PCX_SM_LOCK=2,w2x3k1:PCX_SM_WAVES=2+PCX_SM_PER_CU=3+PCX_SM_LOCK=1,w6x1k3:PCX_SM_WAVES=6+PCX_SM_PER_CU=1+PCX_SM_LOCK=3,w4x2k2:PCX_SM_WAVES=4+PCX_SM_PER_CU=2+PCX_SM_LOCK=2,w8x1k3:PCX_SM_WAVES=8+PCX_SM_PER_CU=1+PCX_SM_LOCK=3,w3x3k1:PCX_SM_WAVES=3+PCX_SM_PER_CU=3+PCX_SM_LOCK=1,w2x4k1:PCX_SM_WAVES=2+PCX_SM_PER_CU=4+PCX_SM_LOCK=1" \
  --out $OUT/ps_sweep.json > $OUT/ps_sweep.txt 2>&1
grep -v "^ *\[" $OUT/ps_sweep.txt | tail -50
