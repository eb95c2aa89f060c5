#!/usr/bin/env python3
"""Times the cropper and post-processor kernels (pcx_crop.hip, pcx_post.hip) on
one GPU and prints, per kernel, the algorithmic bytes it moves, its average
duration (HIP events on the launch stream) and the fraction of the 8 TB/s HBM
roofline.  Writes the table as markdown to stdout (profiles/r02_post_kernels.md
is a copy of one run).

  python tools/post_bench.py [--steps 200]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK = 8000.0  # GB/s


def timed(fn, steps):
  import torch
  for _ in range(10):
    fn()
  torch.cuda.synchronize()
  ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  ev0.record()
  for _ in range(steps):
    fn()
  ev1.record()
  torch.cuda.synchronize()
  return ev0.elapsed_time(ev1) / steps  # ms


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--steps', type=int, default=200)
  args = ap.parse_args()
  import torch
  from pycolab_amd import cropping, rendering
  from pycolab_amd.compiler import GameTemplate
  from pycolab_amd.engine import Engine

  rows = []

  def add(name, what, ms, nbytes):
    rows.append((name, what, ms, nbytes, nbytes / (ms * 1e-3) / 1e9))

  def engine(fixture, batch):
    t = GameTemplate.load(os.path.join(ROOT, 'tests', 'golden', 'templates', fixture + '.npz'))
    eng = Engine.from_template(t, batch=batch, auto_reset=True, seed=1)
    obs = eng.its_showtime()[0]
    eng.step_hashed(7, 0, 20)
    return t, eng, obs

  # croppers: better_scrolly_maze (45x89 board), its own cropper set (better_scrolly_maze.py:237-247)
  t, eng, obs = engine('better_scrolly_maze_L0', 65536)
  P = 1 + len(t.chars)
  for name, cr in (('ScrollingCropper 10x30 on 45x89', cropping.ScrollingCropper(10, 30, ['P'], initial_offset=(0, -4))),
                   ('ScrollingCropper 7x10 padded', cropping.ScrollingCropper(7, 10, ['P'], pad_char='#', scroll_margins=(None, 3))),
                   ('FixedCropper 12x20', cropping.FixedCropper((15, 34), 12, 20))):
    cr.set_engine(eng)
    cr.crop(obs)
    cells = cr.rows * cr.cols
    ms = timed(lambda: cr.crop(obs), args.steps)
    add(name, 'pcx_crop_update + pcx_crop_copy, %d envs, %d planes' % (eng.batch, P), ms,
        eng.batch * (P * cells + P * ((cells + 3) & ~3) + 24))
  eng.close()

  # post-processors on the BASELINE config games
  t, eng, obs = engine('marauders', 32768)
  cells = t.rows * t.cols
  feats = rendering.ObservationToFeatureArray(''.join(chr(c) for c in t.chars))
  feats(obs)
  ms = timed(lambda: feats(obs), args.steps)
  add('ObservationToFeatureArray, all %d layers' % len(t.chars), 'pcx_post_features, marauders %d envs' % eng.batch, ms,
      eng.batch * len(t.chars) * cells * 5)
  hwc = rendering.ObservationToFeatureArray(''.join(chr(c) for c in t.chars), permute=(1, 2, 0))
  hwc(obs)
  ms = timed(lambda: hwc(obs), args.steps)
  add('ObservationToFeatureArray permute=(1,2,0)', 'pcx_post_features (strided stores), marauders %d envs' % eng.batch, ms,
      eng.batch * len(t.chars) * cells * 5)
  rep = rendering.ObservationCharacterRepainter(dict([(b, '^') for b in 'abcd'] + [(b, '|') for b in 'yz']))
  rep(obs)
  depth = len(rep._out_chars)
  ms = timed(lambda: rep(obs), args.steps)
  add('ObservationCharacterRepainter (marauders UI mapping)', 'pcx_post_repaint, marauders %d envs' % eng.batch, ms,
      eng.batch * (cells + (1 + depth) * cells))
  eng.close()

  t, eng, obs = engine('scrolly_maze_L0', 1 << 20)
  cells = t.rows * t.cols
  arr = rendering.ObservationToArray({chr(c): float(i) for i, c in enumerate(t.chars)}, dtype=np.float32)
  arr(obs)
  ms = timed(lambda: arr(obs), args.steps)
  add('ObservationToArray float32 scalars', 'pcx_post_to_array<u32>, scrolly_maze %d envs' % eng.batch, ms,
      eng.batch * cells * 5)
  rgb = rendering.ObservationToArray({chr(c): (i, 2 * i, 255 - i) for i, c in enumerate(t.chars)}, dtype=np.uint8)
  rgb(obs)
  ms = timed(lambda: rgb(obs), args.steps)
  add('ObservationToArray uint8 RGB vectors', 'pcx_post_to_array<u8>, scrolly_maze %d envs' % eng.batch, ms,
      eng.batch * cells * 4)
  feats = rendering.ObservationToFeatureArray(''.join(chr(c) for c in t.chars))
  feats(obs)
  ms = timed(lambda: feats(obs), args.steps)
  add('ObservationToFeatureArray, all %d layers' % len(t.chars), 'pcx_post_features, scrolly_maze %d envs' % eng.batch, ms,
      eng.batch * len(t.chars) * cells * 5)
  eng.close()

  print('| post-processor | kernel, workload | ms | algorithmic MB | GB/s | of 8 TB/s |')
  print('|---|---|---|---|---|---|')
  for name, what, ms, nbytes, gbs in rows:
    print('| %s | %s | %.4f | %.1f | %.0f | %.1f %% |' % (name, what, ms, nbytes / 1e6, gbs, 100 * gbs / HBM_PEAK))


if __name__ == '__main__':
  main()
