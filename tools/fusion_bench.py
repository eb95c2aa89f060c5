#!/usr/bin/env python3
"""Round 3: does fusing pay by itself?  HIP events, one box.
  feature array: step + pcx_post_features (two kernels) vs the epilogue, layers kept (PCX_EPI_SPLIT=0/1 from the
  environment decides whether the cooperative shape splits uint8 / float32 planes between its waves)
  windows only: better_scrolly_maze L0 with the example's three croppers fused, at two batch sizes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pycolab_amd import cropping, rendering
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


def engine(name, batch):
  t = GameTemplate.load(os.path.join(ROOT, 'tests', 'golden', 'templates', name + '.npz'))
  eng = Engine.from_template(t, batch=batch, auto_reset=True, seed=1)
  eng.its_showtime()
  tape = torch.randint(0, t.n_actions, (16, batch), dtype=torch.int32, device='cuda')
  c = [0]
  def one():
    eng.step(tape[c[0] % 16]); c[0] += 1
  return t, eng, one

what = sys.argv[1] if len(sys.argv) > 1 else 'all'
if what in ('all', 'epi', 'sm'):
  for name, batch in ((('scrolly_maze_L0', 1048576),) if what == 'sm' else
                      (('marauders', 32768), ('marauders', 262144), ('hello_world', 65536), ('warehouse_L0', 65536))):
    t, eng, one = engine(name, batch)
    chars = ''.join(chr(c) for c in t.chars)
    obs = eng._result()[0]
    post = rendering.ObservationToFeatureArray(chars)
    post(obs)
    ms_step = sorted(timed(one) for _ in range(3))[1]
    ms_two = sorted(timed(lambda: (one(), post(obs))) for _ in range(3))[1]
    fused = rendering.ObservationToFeatureArray(chars)
    assert fused.fuse_into(eng)
    ms_fused = sorted(timed(one) for _ in range(3))[1]
    print('%-14s %7d envs  two_pass=%s  step %.4f  step+post %.4f  fused %.4f ms' % (name, batch, os.environ.get('PCX_EPI_TWO_PASS', 'auto'), ms_step, ms_two, ms_fused), flush=True)
    eng.close()
if what in ('all', 'hwc'):  # channels last: step + pcx_post_features_hwc vs the epilogue exchanging through LDS
  for name, batch in (('marauders', 32768), ('marauders', 262144), ('hello_world', 65536), ('scrolly_maze_L0', 4096), ('scrolly_maze_L0', 1048576)):
    t, eng, one = engine(name, batch)
    chars = ''.join(chr(c) for c in t.chars)
    obs = eng._result()[0]
    post = rendering.ObservationToFeatureArray(chars, permute=(1, 2, 0))
    post(obs)
    ms_two = sorted(timed(lambda: (one(), post(obs))) for _ in range(3))[1]
    fused = rendering.ObservationToFeatureArray(chars, permute=(1, 2, 0))
    assert fused.fuse_into(eng)
    ms_fused = sorted(timed(one) for _ in range(3))[1]
    print('%-14s %7d envs  channels last: step+post %.4f  fused %.4f ms' % (name, batch, ms_two, ms_fused), flush=True)
    eng.close()
if what in ('all', 'win'):
  for batch in (65536, 262144):
    t, eng, one = engine('better_scrolly_maze_L0', batch)
    crs = [cropping.ScrollingCropper(10, 30, ['P'], initial_offset=(-2, -12)),
           cropping.ScrollingCropper(7, 10, ['c'], pad_char=' ', scroll_margins=(None, 3)),
           cropping.FixedCropper((3, 9), 12, 20, pad_char=' ')]
    P = 1 + len(t.chars)
    win_bytes = batch * sum(P * ((cr.rows * cr.cols + 3) & ~3) + 24 for cr in crs)
    ms_step = sorted(timed(one, 50) for _ in range(3))[1]
    assert cropping.fuse_croppers(eng, crs) is True
    ms_f = sorted(timed(one, 50) for _ in range(3))[1]
    assert cropping.fuse_croppers(eng, crs, only_crops=True) is True
    ms_o = sorted(timed(one, 50) for _ in range(3))[1]
    only_bytes = batch * (32131 - P * 4008) + win_bytes
    print('better_scrolly %7d envs: step %.4f  fused %.4f  windows only %.4f ms = %.1f %% of 8 TB/s' % (
        batch, ms_step, ms_f, ms_o, 100 * only_bytes / (ms_o * 1e-3) / 8e12), flush=True)
    eng.close()
