#!/bin/bash
# round 5, call 16: the launch-shape tuner in pcx_warehouse_step / pcx_hello_world_step
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call16; mkdir -p $OUT
cd $ROOT
echo skip

PCX_DEBUG=16 timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, os
sys.path.insert(0, os.getcwd())
import bench
for game, B in (('warehouse', 131072), ('warehouse', 262144), ('warehouse', 1048576), ('hello_world', 262144), ('hello_world', 1048576)):
  row = bench.measure_config(game, 0, B, 100, 50, 0)
  os.environ[{'warehouse': 'PCX_WM_PW', 'hello_world': 'PCX_HW_PW'}[game]] = '0'
  old = bench.measure_config(game, 0, B, 100, 50, 0)
  del os.environ[{'warehouse': 'PCX_WM_PW', 'hello_world': 'PCX_HW_PW'}[game]]
  print('%-12s %8d  %.4f ms %.3f (shape %d)   round-2 shape %.4f ms %.3f' % (game, B, row['ms_per_step'], row['hbm_frac'], row['launch_shape'], old['ms_per_step'], old['hbm_frac']), flush=True)
PY
