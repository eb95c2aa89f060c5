#!/bin/bash
# round 5, call 8: cooperative instance with compiled-in constants (config 2), block-timed tuner
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call8; mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_persistent_shapes.py tests/test_hip_parity.py -q -m gpu -x -k "cooperative or several_steps or compiled_in or headline or hashed_actions" > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt
tail -4 $OUT/pytest.txt
for B in 65536 131072 262144 524288 1048576; do
  PCX_DEBUG=16 timeout 300 python bench.py --batch $B --steps 100 --warmup 40 --repeats 3 --no-other-configs --no-cpu-baseline 2> $OUT/tune_$B.err | python -c "
import json,sys
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); print($B, 'ms', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'achievable', d['roofline'].get('achievable',{}).get('GBps'))"
  grep "pcx scrolly" $OUT/tune_$B.err | tail -1
done
for baked in 1 0; do
  echo "== config 2 (4,096 environments), PCX_SM_BAKED=$baked"
  PCX_SM_BAKED=$baked timeout 300 python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import bench
for row in (bench.measure_config('scrolly_maze', 0, 4096, 400, 40, 0), bench.measure_step_n('scrolly_maze', 0, 4096, 1000, 0),
            bench.measure_config('scrolly_maze', 0, 16384, 400, 40, 0), bench.measure_config('scrolly_maze', 0, 32768, 400, 40, 0)):
  print('  %-60s %.4f ms  %.3f of 8 TB/s  shape %d' % (row['workload'][:60], row['ms_per_step'], row['hbm_frac'], row['launch_shape']))
PY
done
