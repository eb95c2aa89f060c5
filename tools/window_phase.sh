#!/bin/bash
# Where the fused croppers' time goes (better_scrolly_maze L0, 65,536 envs): logic only (PCX_DEBUG=2 skips the
# streaming phases), full step, fused, windows only.
for D in 0 2; do PCX_DEBUG=$D python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from pycolab_amd import cropping
from pycolab_amd.compiler import GameTemplate
from pycolab_amd.engine import Engine
def timed(fn, steps=100):
  for _ in range(10): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(steps): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / steps
t = GameTemplate.load('tests/golden/templates/better_scrolly_maze_L0.npz')
for debug in (os.environ['PCX_DEBUG'],):
  eng = Engine.from_template(t, batch=65536, auto_reset=True, seed=1)
  eng.its_showtime()
  tape = torch.randint(0, 5, (16, eng.batch), dtype=torch.int32, device='cuda')
  c = [0]
  def one():
    eng.step(tape[c[0] % 16]); c[0] += 1
  print('PCX_DEBUG=%s plain      %.4f ms' % (debug, timed(one)))
  crs = [cropping.ScrollingCropper(10, 30, ['P'], initial_offset=(-2, -12)),
         cropping.ScrollingCropper(7, 10, ['c'], pad_char=' ', scroll_margins=(None, 3)),
         cropping.FixedCropper((3, 9), 12, 20, pad_char=' ')]
  assert cropping.fuse_croppers(eng, crs)
  print('PCX_DEBUG=%s fused      %.4f ms' % (debug, timed(one)))
  assert cropping.fuse_croppers(eng, crs, only_crops=True)
  print('PCX_DEBUG=%s only_crops %.4f ms' % (debug, timed(one)))
  assert cropping.fuse_croppers(eng, crs[:1], only_crops=True)
  print('PCX_DEBUG=%s only the 10x30 window %.4f ms' % (debug, timed(one)))
  eng.close()
PY
done
