#!/usr/bin/env python3
"""pcx_generic_step (the table-driven kernel) on the workloads VERDICT r2 names, HIP events.
  python tools/generic_timing.py            every case forced through the table-driven kernel
  PCX_FORCE_GENERIC=0 python tools/...      the same cases through whatever kernel the engine picks"""
import os, sys
os.environ.setdefault('PCX_FORCE_GENERIC', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pycolab_amd import _native as N
from pycolab_amd.compiler import GameTemplate
from pycolab_amd.engine import Engine

CASES = [('warehouse_L0', 262144), ('marauders_custom_A', 32768), ('marauders', 32768), ('walkers_scroll_groups', 262144),
         ('directives_z_order', 262144), ('walkers_room', 262144), ('hello_world', 262144), ('warehouse_L0_unoccluded', 262144),
         ('better_scrolly_custom_B', 262144), ('marauders_unoccluded', 32768)]
if len(sys.argv) > 1:
  CASES = [(a.split(':')[0], int(a.split(':')[1])) for a in sys.argv[1:]]


def timed(fn, steps=100):
  for _ in range(10): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(steps): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / steps


for name, batch in CASES:
  t = GameTemplate.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'templates', name + '.npz'))
  eng = Engine.from_template(t, batch=batch, auto_reset=True, seed=1)
  eng.its_showtime()
  tape = torch.randint(0, max(1, t.n_actions), (16, batch), dtype=torch.int32, device='cuda')
  c = [0]
  def one():
    eng.step(tape[c[0] % 16]); c[0] += 1
  ms = sorted(timed(one) for _ in range(3))[1]
  bps = int(N.lib().pcx_engine_bytes_per_step(eng._native))
  print('%-26s %8d envs  %-24s %.4f ms  %5.1f %% of 8 TB/s  (%d B/env-step, %dx%d, %d chars)' % (
      name, batch, N.lib().pcx_engine_kernel_name(eng._native).decode(), ms, 100 * bps * batch / (ms * 1e-3) / 8e12, bps,
      t.rows, t.cols, len(t.chars)), flush=True)
  eng.close()
