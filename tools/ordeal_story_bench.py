#!/usr/bin/env python3
"""examples/ordeal.py as ONE batched Story (every environment in its own chapter), at scale: the 16 recorded stories of
tests/golden/traces/ordeal_story.npz tiled over the batch -- the game is deterministic, so every copy must show the
recorded boards, rewards and chapters -- timed per play() with the checks outside the timed region.

  python tools/ordeal_story_bench.py [--batch 16384] [--steps 320]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--batch', type=int, default=16384)
  ap.add_argument('--steps', type=int, default=320)
  args = ap.parse_args()
  import torch
  from pycolab_amd import cropping, storytelling
  from pycolab_amd.compiler import GameTemplate
  from pycolab_amd.engine import Engine
  tr = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'traces', 'ordeal_story.npz')))
  T, E = tr['actions'].shape
  T = min(T, args.steps)
  B = args.batch
  keys = ('castle', 'cavern', 'kansas')
  load = lambda k: GameTemplate.load(os.path.join(ROOT, 'tests', 'golden', 'templates', 'ordeal_%s.npz' % k))
  chapters = {k: (lambda k=k: Engine.from_template(load(k), batch=B)) for k in keys}
  story = storytelling.Story(chapters=chapters, croppers=dict(castle=None, cavern=None, kansas=cropping.ScrollingCropper(
      rows=8, cols=15, to_track='P', scroll_margins=(2, 3))), first_chapter='kansas', auto_reset=True)
  tile = np.arange(B) % E
  actions = torch.from_numpy(np.ascontiguousarray(tr['actions'][:, tile])).cuda()
  bad = 0

  def check(result, row):
    obs, reward, discount = result
    ok = (np.array_equal(obs.board.cpu().numpy(), tr['boards'][row][tile]) and np.array_equal(reward, tr['reward'][row][tile]) and
          np.array_equal(discount, tr['discount'][row][tile]) and np.array_equal(story.reward_set, tr['reward_set'][row][tile].astype(bool)) and
          story.this_chapter == [None if c < 0 else keys[c] for c in tr['chapter'][row][tile]])
    return 0 if ok else 1
  bad += check(story.its_showtime(), 0)
  spent, changes = 0.0, 0
  for t in range(T):
    before = np.array([-1 if c is None else keys.index(c) for c in story.this_chapter]) if t % 16 == 0 else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    result = story.play(actions[t])
    torch.cuda.synchronize()
    spent += time.perf_counter() - t0
    if t % 16 == 15 or t == T - 1:
      bad += check(result, t + 1)
  changes = int((np.diff(tr['chapter'][:T + 1][:, tile].astype(np.int32), axis=0) != 0).sum())
  print('ordeal Story, %d environments x %d steps (the 16 recorded stories tiled): %.3f ms per play(), %.2f M env-steps/s; '
        '%d chapter changes / story ends in the run; rows checked against the reference trace: %s' % (
            B, T, 1e3 * spent / T, B * T / spent / 1e6, changes, 'ALL EQUAL' if bad == 0 else '%d MISMATCHES' % bad))
  story.close()
  cpu_reference(tr)
  return 1 if bad else 0


def cpu_reference(tr, seconds=5.0):
  """The reference's own ordeal Story on one host core, replaying the recorded tapes (TEST INFRASTRUCTURE: oracle/_ref)."""
  import collections
  import collections.abc
  import importlib
  import warnings
  from oracle import ref_live
  path = ref_live.reference_path()
  if path is None:
    print('the reference is not on this machine: no CPU figure')
    return
  sys.path.insert(0, path)
  warnings.filterwarnings('ignore')
  for name in ('Mapping', 'Sequence'):
    if not hasattr(collections, name):
      setattr(collections, name, getattr(collections.abc, name))
  ref = importlib.import_module('pycolab.examples.ordeal')
  T, E = tr['actions'].shape
  steps, t0 = 0, time.perf_counter()
  while time.perf_counter() - t0 < seconds:
    for e in range(E):
      story = ref.make_game()
      story.its_showtime()
      for t in range(T):
        if story.game_over:
          story = ref.make_game()
          story.its_showtime()
        else:
          story.play(int(tr['actions'][t, e]))
        steps += 1
      if time.perf_counter() - t0 >= seconds:
        break
  dt = time.perf_counter() - t0
  print('the reference (pycolab.examples.ordeal, one host core, the same tapes): %.1f k env-steps/s' % (steps / dt / 1e3))


if __name__ == '__main__':
  sys.exit(main())
