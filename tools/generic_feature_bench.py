#!/usr/bin/env python3
"""Round 4: the table-driven kernel's feature-array epilogue (pcx_generic_step writes ObservationToFeatureArray in its
default axis order from its render loop).  HIP events, one box: step alone, step + stand-alone pcx_post kernel, and the
fused step with the uint8 planes kept / the layers dropped / every plane dropped; % of 8 TB/s on the bytes each writes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pycolab_amd import _native as N
from pycolab_amd import rendering
from pycolab_amd.compiler import GameTemplate
from pycolab_amd.engine import Engine
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [('walkers_scroll_groups', 262144), ('marauders_custom_A', 32768), ('walkers_room', 262144), ('directives_z_order', 262144)]
if len(sys.argv) > 1:
  CASES = [(a.split(':')[0], int(a.split(':')[1])) for a in sys.argv[1:]]


def timed(fn, steps=100):
  for _ in range(30): fn()  # (the engine settles its waves per workgroup on its first launches)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(steps): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / steps


def engine(name, batch):
  t = GameTemplate.load(os.path.join(ROOT, 'tests', 'golden', 'templates', name + '.npz'))
  eng = Engine.from_template(t, batch=batch, auto_reset=True, seed=1)
  eng.its_showtime()
  tape = torch.randint(0, max(1, t.n_actions), (16, batch), dtype=torch.int32, device='cuda')
  c = [0]
  def one():
    eng.step(tape[c[0] % 16]); c[0] += 1
  return t, eng, one


for name, batch in CASES:
  t, eng, one = engine(name, batch)
  chars = ''.join(chr(c) for c in t.chars)
  cells, L = t.rows * t.cols, len(t.chars)
  obs = eng._result()[0]
  post = rendering.ObservationToFeatureArray(chars)
  post(obs)
  ms_step = sorted(timed(one) for _ in range(3))[1]
  ms_two = sorted(timed(lambda: (one(), post(obs))) for _ in range(3))[1]
  eng.close()
  state = int(N.lib().pcx_engine_bytes_per_step(eng._native)) if eng._native else 0
  row = '%-22s %7d envs  step %.4f  step+post %.4f ms' % (name, batch, ms_step, ms_two)
  for label, kw, planes in (('kept', {}, (1 + L) * cells), ('skip_layers', dict(skip_layers=True), cells), ('skip_board', dict(skip_board=True), 0)):
    t, eng, one = engine(name, batch)
    fused = rendering.ObservationToFeatureArray(chars)
    assert fused.fuse_into(eng, **kw)
    ms = sorted(timed(one) for _ in range(3))[1]
    own = planes + 4 * L * cells  # bytes written per environment: uint8 planes kept + the float32 stack
    row += '  | fused %s %.4f ms (%.0f %% of 8 TB/s on %d B/env)' % (label, ms, 100 * own * batch / (ms * 1e-3) / 8e12, own)
    eng.close()
  print(row, flush=True)
