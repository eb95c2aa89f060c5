#!/usr/bin/env python3
"""Crop -> post-process in one launch: better_scrolly_maze level 0, 65,536 environments, the example's
egocentric 10 x 30 window (better_scrolly_maze.py:237-247) as an eight-layer float32 stack (rendering.py:545-661).
ms per step, HIP events, median of three; the last column is the fraction of 8 TB/s that the launch's OWN algorithmic
bytes come to (what it must read and write for what it returns)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pycolab_amd import cropping, rendering
from pycolab_amd.compiler import GameTemplate
from pycolab_amd.engine import Engine
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t = GameTemplate.load(os.path.join(ROOT, 'tests/golden/templates/better_scrolly_maze_L0.npz'))
LAYERS = 'P@#abc +'


def timed(fn, steps=60):
  for _ in range(8): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(steps): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / steps


for B in [int(x) for x in (sys.argv[1:] or ['65536', '262144'])]:
  cells, wcells, L, D = t.rows * t.cols, 300, len(t.chars), len(LAYERS)
  state = 4 + 2 * 36 + 15
  rows = []
  for mode in ('step + stand-alone cropper + stand-alone features (3 kernels)', 'fused cropper, stand-alone features',
               'fused cropper + fused window stack', '... window uint8 planes dropped', '... and full-board planes dropped (only_crops)'):
    eng = Engine.from_template(t, batch=B, auto_reset=True, seed=1)
    cr = cropping.ScrollingCropper(rows=10, cols=30, to_track=['P'], scroll_margins=(2, 3), initial_offset=(-3, -9))
    cr.set_engine(eng)
    eng.its_showtime()
    tape = torch.randint(0, 5, (16, B), dtype=torch.int32, device='cuda')
    conv = rendering.ObservationToFeatureArray(LAYERS)
    c = [0]
    def one():
      eng.step(tape[c[0] % 16]); c[0] += 1
      return conv(cr.crop(None))
    full = (1 + L) * cells
    if mode.startswith('step'):
      algo = state + full + 2 * (1 + L) * wcells + (1 + L) * wcells + D * wcells * 4
    elif mode.startswith('fused cropper,'):
      assert cropping.fuse_croppers(eng, [cr])
      algo = state + full + (1 + L) * wcells + (1 + L) * wcells + D * wcells * 4
    else:
      only = 'only_crops' in mode
      assert cropping.fuse_croppers(eng, [cr], only_crops=only)
      drop = mode.startswith('...')
      assert conv.fuse_into(eng, source=cr, skip_board=drop)
      algo = state + (0 if only else full) + (0 if drop else (1 + L) * wcells) + D * wcells * 4
    ms = sorted(timed(one) for _ in range(3))[1]
    rows.append((mode, ms, algo * B / (ms * 1e-3) / 8e12))
    eng.close()
  for mode, ms, frac in rows:
    print('%8d envs  %-66s %.4f ms   %.3f of 8 TB/s on its own bytes' % (B, mode, ms, frac), flush=True)
