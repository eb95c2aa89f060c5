#!/usr/bin/env python3
"""Several steps per launch against one launch per step at small batches: scrolly_maze level 0, us per
step by HIP events around 2,000 steps -- T single `step()` calls, `step_n(tape)` through the cooperative instance that
walks the launch's steps (round 4, launch shape 12), the same through round 1's multi-step instance (PCX_TFUSE_OLD=1,
shape 11), and `step_hashed`.  One box, one process, alternating."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pycolab_amd import _native as N
from pycolab_amd.compiler import GameTemplate
from pycolab_amd.engine import Engine

t = GameTemplate.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/templates/scrolly_maze_L0.npz'))
T = 2000


def timed(fn):
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / T * 1e3


print('%8s %14s %14s %14s %14s' % ('batch', 'T x step()', 'step_n new', 'step_n round-1', 'step_hashed new'))
for B in [int(x) for x in (sys.argv[1:] or ['256', '4096', '16384', '65536'])]:
  eng = Engine.from_template(t, batch=B, device=0, auto_reset=True)
  eng.its_showtime()
  tape = torch.randint(0, 5, (T, B), dtype=torch.int32, device='cuda')
  ptr = tape.data_ptr()
  lib, stream = N.lib(), None
  import ctypes
  from pycolab_amd import device as dev
  stream = dev.current_stream(0)
  step_n = lambda: N.check(lib.pcx_engine_step_n(eng._native, ptr, T, 1, stream))
  hashed = lambda: N.check(lib.pcx_engine_step_hashed(eng._native, 1, 0, 0, T, 1, stream))
  rows = {'single': [], 'new': [], 'old': [], 'hashed': []}
  shapes = {}
  for rep in range(3):
    os.environ.pop('PCX_TFUSE_OLD', None)
    for i in range(50):
      eng.step(tape[i])
    rows['single'].append(timed(lambda: [eng.step(tape[i]) for i in range(T)]))
    step_n()
    rows['new'].append(timed(step_n))
    shapes['new'] = int(lib.pcx_engine_launch_shape(eng._native))
    rows['hashed'].append(timed(hashed))
    os.environ['PCX_TFUSE_OLD'] = '1'
    step_n()
    rows['old'].append(timed(step_n))
    shapes['old'] = int(lib.pcx_engine_launch_shape(eng._native))
  os.environ.pop('PCX_TFUSE_OLD', None)
  med = lambda xs: sorted(xs)[len(xs) // 2]
  print('%8d %11.2f us %11.2f us %11.2f us %11.2f us   (shapes %s)' % (B, med(rows['single']), med(rows['new']), med(rows['old']), med(rows['hashed']), shapes), flush=True)
  eng.check_errors()
  eng.close()
