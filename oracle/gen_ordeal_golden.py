#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Golden trace of the reference's examples/ordeal.py (SURVEY.md section 8 f-4's cited game).

Runs google-deepmind/pycolab's own `ordeal.make_game()` -- a `storytelling.Story` of three games whose entities add float
rewards (ordeal.py:123, 187-190), keep 'has_sword' / 'last_position' in the Plot, read the_plot.this_chapter /
prior_chapter and assign the_plot.next_chapter (:113-269) -- on goal-directed action tapes with noise
(oracle/ordeal_story.py tape_action: decided from where the reference's player stands; the actions are recorded), and
records per step the (cropped) board, reward (+ "is None" flag), discount, game_over, the chapter, and the two Plot
entries.  A story that ends is replaced by a new one at the next step.

Run here (CPU container):  python oracle/gen_ordeal_golden.py
Output: tests/golden/traces/ordeal_story.npz
"""
import collections
import collections.abc
import os
import sys
import warnings

import numpy as np

for _name in ('Mapping', 'Sequence'):  # storytelling.py uses collections.Mapping, gone since Python 3.10
  if not hasattr(collections, _name):
    setattr(collections, _name, getattr(collections.abc, _name))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('PCX_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)
sys.dont_write_bytecode = True
warnings.filterwarnings('ignore', category=DeprecationWarning)

from pycolab.examples import ordeal as ref_ordeal  # noqa: E402

from oracle import ordeal_story  # noqa: E402


def main():
  E, T = 16, 320
  actions = np.zeros((T, E), np.int32)
  cols = collections.defaultdict(list)
  seen = collections.Counter()
  for e in range(E):
    rng = np.random.RandomState(7700 + e)
    style = ordeal_story.style_of(e)
    story = ref_ordeal.make_game()
    rec = []

    def note(obs, r, d, fresh):
      plot = story.the_plot
      over = story.game_over
      lp = plot.get('last_position', (-1, -1))
      assert r is None or isinstance(r, float), r
      rec.append((obs.board.copy(), 0.0 if r is None else r, 0 if r is None else 1, float(d), int(over),
                  -1 if over else ordeal_story.KEYS.index(plot.this_chapter), int(fresh),
                  int(bool(plot.get('has_sword'))), int(lp[0]), int(lp[1])))
      seen['%s%s' % (plot.this_chapter, '' if r is None else ' %+g' % r)] += 1
    note(*story.its_showtime(), 1)
    for t in range(T):
      if story.game_over:
        story = ref_ordeal.make_game()
        note(*story.its_showtime(), 1)
        continue
      p = story.current_game.things['P'].position
      a = ordeal_story.tape_action(rng, style, story.the_plot.this_chapter, p.row, p.col, bool(story.the_plot.get('has_sword')), t)
      actions[t, e] = a
      note(*story.play(a), 0)
    for i, name in enumerate(('boards', 'reward', 'reward_set', 'discount', 'done', 'chapter', 'fresh', 'has_sword', 'last_row', 'last_col')):
      cols[name].append([x[i] for x in rec])
  sw = lambda x, dt: np.ascontiguousarray(np.swapaxes(np.array(x, dtype=dt), 0, 1))
  out_root = os.environ.get('PCX_GOLDEN_OUT') or os.path.join(ROOT, 'tests', 'golden')
  os.makedirs(os.path.join(out_root, 'traces'), exist_ok=True)
  path = os.path.join(out_root, 'traces', 'ordeal_story.npz')
  dts = dict(boards=np.uint8, reward=np.float32, reward_set=np.uint8, discount=np.float32, done=np.uint8, chapter=np.int8,
             fresh=np.uint8, has_sword=np.uint8, last_row=np.int16, last_col=np.int16)
  np.savez_compressed(path, actions=actions, **{k: sw(v, dts[k]) for k, v in cols.items()})
  print('wrote %s (%d bytes): %s' % (path, os.path.getsize(path), dict(sorted(seen.items()))))


if __name__ == '__main__':
  main()
