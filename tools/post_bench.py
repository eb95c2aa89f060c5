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
  # the example's own cropper set (better_scrolly_maze.py:237-247) on every observation: as their own kernels
  # after the step, fused into the step kernel, and fused with the full-board planes no longer written
  for cr in list(eng._croppers):
    cr.set_engine(None)  # detach the stand-alone croppers timed above
  def example_croppers():
    return [cropping.ScrollingCropper(10, 30, ['P'], initial_offset=(-2, -12)),
            cropping.ScrollingCropper(7, 10, ['c'], pad_char=' ', scroll_margins=(None, 3)),
            cropping.FixedCropper((3, 9), 12, 20, pad_char=' ')]
  tape = torch.randint(0, 5, (16, eng.batch), dtype=torch.int32, device='cuda')
  counter = [0]

  def one_step():
    eng.step(tape[counter[0] % 16]); counter[0] += 1
  crs = example_croppers()
  for cr in crs:
    cr.set_engine(eng)
    cr.crop(obs)
  win_bytes = eng.batch * sum(P * ((cr.rows * cr.cols + 3) & ~3) + 24 for cr in crs)
  step_bytes = eng.batch * 32131
  ms_step = timed(one_step, args.steps)
  ms_sep = timed(lambda: (one_step(), [cr.crop(obs) for cr in crs]), args.steps)
  assert cropping.fuse_croppers(eng, crs) is True
  ms_fused = timed(one_step, args.steps)
  assert cropping.fuse_croppers(eng, crs, only_crops=True) is True
  ms_only = timed(one_step, args.steps)
  add('play() alone, better_scrolly_maze 45x89', 'pcx_better_scrolly_step, %d envs' % eng.batch, ms_step, step_bytes)
  add('play() + the example\'s three croppers (seven kernels)', 'pcx_better_scrolly_step + 3 x (pcx_crop_update + pcx_crop_copy)',
      ms_sep, step_bytes + win_bytes + eng.batch * sum(P * cr.rows * cr.cols for cr in crs))
  add('play() with the three croppers fused', 'pcx_better_scrolly_step', ms_fused, step_bytes + win_bytes)
  add('... fused, windows only (full-board planes not written)', 'pcx_better_scrolly_step', ms_only,
      step_bytes - eng.batch * P * 4008 + win_bytes)
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
  add('ObservationToFeatureArray permute=(1,2,0)', 'pcx_post_features_hwc, marauders %d envs' % eng.batch, ms,
      eng.batch * len(t.chars) * cells * 5)
  # the same feature array as an epilogue of the step kernel: step alone, step + separate kernel, fused step
  tape = torch.randint(0, 4, (64, eng.batch), dtype=torch.int32, device='cuda')
  counter = [0]

  def one_step():
    eng.step(tape[counter[0] % 64]); counter[0] += 1
  ms_step = timed(one_step, args.steps)
  ms_both = timed(lambda: (one_step(), feats(obs)), args.steps)
  fused = rendering.ObservationToFeatureArray(''.join(chr(c) for c in t.chars))
  assert fused.fuse_into(eng)
  ms_fused = timed(one_step, args.steps)
  assert fused.fuse_into(eng, skip_layers=True)
  ms_fused_only = timed(one_step, args.steps)
  step_bytes = eng.batch * 7283
  feat_bytes = eng.batch * len(t.chars) * cells * 4
  add('play() alone', 'pcx_marauders_step, %d envs' % eng.batch, ms_step, step_bytes)
  add('play() + ObservationToFeatureArray (two kernels)', 'pcx_marauders_step + pcx_post_features', ms_both,
      step_bytes + feat_bytes + eng.batch * len(t.chars) * cells)
  add('play() with the feature array fused (epilogue)', 'pcx_marauders_step', ms_fused, step_bytes + feat_bytes)
  add('... fused, uint8 layer planes skipped', 'pcx_marauders_step', ms_fused_only,
      step_bytes + feat_bytes - eng.batch * len(t.chars) * cells)
  N_ = __import__('pycolab_amd._native', fromlist=['x'])
  N_.check(N_.lib().pcx_engine_set_epilogue(eng._native, None))
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
  tape = torch.randint(0, 5, (16, eng.batch), dtype=torch.int32, device='cuda')
  counter = [0]

  def one_step():
    eng.step(tape[counter[0] % 16]); counter[0] += 1
  ms_step = timed(one_step, 50)
  ms_both = timed(lambda: (one_step(), feats(obs)), 50)
  fused = rendering.ObservationToFeatureArray(''.join(chr(c) for c in t.chars))
  assert fused.fuse_into(eng)
  ms_fused = timed(one_step, 50)
  assert fused.fuse_into(eng, skip_layers=True)
  ms_fused_only = timed(one_step, 50)
  step_bytes, feat_bytes, lay_bytes = eng.batch * 2819, eng.batch * len(t.chars) * cells * 4, eng.batch * len(t.chars) * cells
  add('play() alone', 'pcx_scrolly_maze_step, %d envs' % eng.batch, ms_step, step_bytes)
  add('play() + ObservationToFeatureArray (two kernels)', 'pcx_scrolly_maze_step + pcx_post_features', ms_both, step_bytes + feat_bytes + lay_bytes)
  add('play() with the feature array fused (epilogue)', 'pcx_scrolly_maze_step', ms_fused, step_bytes + feat_bytes)
  add('... fused, uint8 layer planes skipped', 'pcx_scrolly_maze_step', ms_fused_only, step_bytes + feat_bytes - lay_bytes)
  eng.close()

  print('| post-processor | kernel, workload | ms | algorithmic MB | GB/s | of 8 TB/s |')
  print('|---|---|---|---|---|---|')
  for name, what, ms, nbytes, gbs in rows:
    print('| %s | %s | %.4f | %.1f | %.0f | %.1f %% |' % (name, what, ms, nbytes / 1e6, gbs, 100 * gbs / HBM_PEAK))


if __name__ == '__main__':
  main()
