#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Golden trace of the reference's `storytelling.Story`.

Runs google-deepmind/pycolab's own Story (pycolab/storytelling.py:36-475) over
three chapters built from the reference's test entities
(oracle/directive_scenarios.py: STORY), on seeded action tapes, and records per
step the board, reward (+ "is None" flag), discount, game_over and the chapter
the story is in.  A story that ends is replaced by a new one at the next step
(like the engine traces).  storytelling.py uses `collections.Mapping`, which
Python 3.10 no longer has: the aliases are restored before importing it.

Run here (CPU container):  python oracle/gen_story_golden.py
Output: tests/golden/traces/story_three_chapters.npz
"""
import collections
import collections.abc
import os
import sys
import warnings

import numpy as np

for _name in ('Mapping', 'Sequence'):
  if not hasattr(collections, _name):
    setattr(collections, _name, getattr(collections.abc, _name))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('PCX_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)
sys.dont_write_bytecode = True
warnings.filterwarnings('ignore', category=DeprecationWarning)

from pycolab import ascii_art as ref_art  # noqa: E402
from pycolab import storytelling as ref_story  # noqa: E402
from pycolab.tests import test_things as tt  # noqa: E402

from oracle import directive_scenarios as ds  # noqa: E402


def main():
  generate('story_three_chapters', ds.STORY, 9000)
  generate('story_entity_chapters', ds.STORY_JUMPS, 9100)  # the entities name the next chapter (plot.py:299-324)


def generate(name, specs, seed0):
  def make_story():
    return ref_story.Story([lambda spec=spec: ds.build_reference(spec, ref_art, tt) for spec in specs])
  E, T = 16, 96
  chars = sorted(set('.').union(*[set(spec['entities']) for spec in specs]))
  actions = np.zeros((T, E), np.int32)
  boards, rewards, rsets, discounts, dones, chapters, restarted = [], [], [], [], [], [], []
  for e in range(E):
    rng = np.random.RandomState(seed0 + e)
    actions[:, e] = ds.story_tape(rng, T)
    story = make_story()
    rec = []

    def note(obs, r, d, fresh):
      rec.append((obs.board.copy(), 0 if r is None else int(r), 0 if r is None else 1, float(d), int(story.game_over),
                  -1 if story.game_over else int(story.the_plot.this_chapter), int(fresh)))
    obs, r, d = story.its_showtime()
    note(obs, r, d, 1)
    for t in range(T):
      if story.game_over:
        story = make_story()
        obs, r, d = story.its_showtime()
        note(obs, r, d, 1)
        continue
      a = int(actions[t, e])
      spec = specs[story.the_plot.this_chapter]
      ds.inject(spec, story.current_game, a, tt)
      obs, r, d = story.play(ds.reference_action(spec, a))
      note(obs, r, d, 0)
    boards.append([x[0] for x in rec]); rewards.append([x[1] for x in rec]); rsets.append([x[2] for x in rec])
    discounts.append([x[3] for x in rec]); dones.append([x[4] for x in rec]); chapters.append([x[5] for x in rec])
    restarted.append([x[6] for x in rec])
  sw = lambda x, dt: np.ascontiguousarray(np.swapaxes(np.array(x, dtype=dt), 0, 1))
  out_root = os.environ.get('PCX_GOLDEN_OUT') or os.path.join(ROOT, 'tests', 'golden')
  os.makedirs(os.path.join(out_root, 'traces'), exist_ok=True)
  path = os.path.join(out_root, 'traces', name + '.npz')
  np.savez_compressed(path, actions=actions, boards=sw(boards, np.uint8), reward=sw(rewards, np.int32),
                      reward_set=sw(rsets, np.uint8), discount=sw(discounts, np.float32), done=sw(dones, np.uint8),
                      chapter=sw(chapters, np.int8), fresh=sw(restarted, np.uint8),
                      chars=np.array([ord(c) for c in chars], np.uint8))
  print('wrote %s: stories ended %d times, chapter visits %s' % (
      path, int(np.array(dones).sum()), np.bincount(np.array(chapters).ravel() + 1)))


if __name__ == '__main__':
  main()
