#!/usr/bin/env python3
"""ObservationCharacterRepainter: step + pcx_post_repaint (two kernels) vs the fused epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pycolab_amd import rendering
from pycolab_amd.compiler import GameTemplate
from pycolab_amd.engine import Engine
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def timed(fn, steps=100):
  for _ in range(10): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(steps): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / steps


for name, batch in (('marauders', 32768), ('scrolly_maze_L0', 4096), ('scrolly_maze_L0', 1048576), ('hello_world', 262144)):
  t = GameTemplate.load(os.path.join(ROOT, 'tests', 'golden', 'templates', name + '.npz'))
  chars = [chr(c) for c in t.chars]
  mapping = {chars[1]: chars[0], chars[-1]: chars[-2]}  # two pairs of characters merge (as the examples' UI mappings do)
  res = []
  for mode in ('two kernels', 'fused', 'fused, skip_board'):
    eng = Engine.from_template(t, batch=batch, auto_reset=True, seed=1)
    eng.its_showtime()
    tape = torch.randint(0, t.n_actions, (16, batch), dtype=torch.int32, device='cuda')
    c = [0]
    def one():
      eng.step(tape[c[0] % 16]); c[0] += 1
    conv = rendering.ObservationCharacterRepainter(mapping)
    obs = eng._result()[0]
    if mode == 'two kernels':
      conv(obs)
      fn = lambda: (one(), conv(obs))
    else:
      assert conv.fuse_into(eng, skip_board=mode != 'fused', force=True)
      fn = one
    res.append(sorted(timed(fn, 50 if batch > 500000 else 100) for _ in range(3))[1])
    eng.close()
  print('%-16s %8d envs  repainter  step+post %.4f  fused %.4f  fused, repainted planes only %.4f ms' % ((name, batch) + tuple(res)), flush=True)
