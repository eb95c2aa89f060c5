#!/usr/bin/env python3
"""Feature epilogue with skip_layers (the uint8 layer planes are not written): one sweep vs two (PCX_EPI_TWO_PASS)."""
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


for name, batch in (('marauders', 32768), ('marauders', 262144), ('hello_world', 65536), ('scrolly_maze_L0', 1048576)):
  for permute in (None, (1, 2, 0)):
    t = GameTemplate.load(os.path.join(ROOT, 'tests', 'golden', 'templates', name + '.npz'))
    eng = Engine.from_template(t, batch=batch, auto_reset=True, seed=1)
    eng.its_showtime()
    tape = torch.randint(0, t.n_actions, (16, batch), dtype=torch.int32, device='cuda')
    c = [0]
    def one():
      eng.step(tape[c[0] % 16]); c[0] += 1
    fused = rendering.ObservationToFeatureArray(''.join(chr(x) for x in t.chars), permute=permute)
    assert fused.fuse_into(eng, skip_layers=True)
    ms = sorted(timed(one) for _ in range(3))[1]
    print('%-16s %8d envs  %-13s skip_layers  PCX_EPI_TWO_PASS=%s  %.4f ms' % (
        name, batch, 'channels last' if permute else 'planar', os.environ.get('PCX_EPI_TWO_PASS', 'auto'), ms), flush=True)
    eng.close()
