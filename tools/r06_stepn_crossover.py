#!/usr/bin/env python3
"""Engine.step_n (several steps per launch: pcx_scrolly_maze_step's multi-step persistent shape 13) against single-step
launches (shape 5: tickets + stealing) by batch size: where does walking the steps inside a launch stop paying?
  python tools/r06_stepn_crossover.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

for batch in (65536, 131072, 262144, 393216, 524288, 786432, 1048576):
  steps = 256 if batch <= 262144 else 128
  fused = bench.measure_step_n('scrolly_maze', 0, batch, steps, 0)
  os.environ['PCX_FUSE_STEPS'] = '0'
  try:
    single_n = bench.measure_step_n('scrolly_maze', 0, batch, steps, 0)
  finally:
    del os.environ['PCX_FUSE_STEPS']
  single = bench.measure_config('scrolly_maze', 0, batch, steps, 20, 0)
  print('%8d envs: step_n fused %.4f ms (shape %s)  step_n as single launches %.4f (shape %s)  Engine.step %.4f (shape %s)' % (
      batch, fused['ms_per_step'], fused['launch_shape'], single_n['ms_per_step'], single_n['launch_shape'], single['ms_per_step'], single['launch_shape']))
  sys.stdout.flush()
