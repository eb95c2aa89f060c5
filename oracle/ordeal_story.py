"""TEST INFRASTRUCTURE.  examples/ordeal.py -- the Story SURVEY.md section 8 f-4 cites -- for the parity tests.

* `tape_action`: a goal-directed walk with noise (to the castle without the sword / to the cavern for the sword and then
  to the castle / wandering / quitting), decided from where the REFERENCE's player stands: random actions alone never
  leave Kansas.  The actions it produced are what the fixtures record; the tests replay them.
* `OracleOrdeal`: storytelling.Story (storytelling.py:172-283, 391-470) over one-environment oracle engines built from
  the three chapter templates (tests/golden/templates/ordeal_*.npz: the unchanged example file compiled by
  pycolab_amd.compiler), the Kansas chapter behind the example's ScrollingCropper (ordeal.py:103-105), the Plot entries
  of the example's entities handed from game to game as plot words (include/pcx.h PCX_PLOT_WORDS).
"""
import os

import numpy as np

from oracle import binding
from pycolab_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ('castle', 'cavern', 'kansas')  # chapter codes: the keys sorted (include/pcx.h PCX_PROG_OD_PLAYER)
STYLES = ('castle', 'sword', 'wander', 'quit')
N_ACTIONS = 9  # 0 N, 1 S, 2 W, 3 E, 4 quit; 5..8 do nothing (ordeal.py:216-246: no clause matches)


def style_of(story_index):
  return STYLES[story_index % 4] if story_index % 7 != 6 else 'sword'


def tape_action(rng, style, chapter, row, col, has_sword, t):
  """The next action of a story of `style` whose player stands at (row, col) of `chapter`."""
  if rng.rand() < 0.12:
    return int(rng.randint(0, 4))
  if rng.rand() < 0.03:
    return int(rng.randint(5, N_ACTIONS))
  if style == 'quit' and t > 0 and rng.rand() < 0.05:
    return 4
  if style == 'wander':
    return int(rng.randint(0, 4))
  goal_castle = style in ('castle', 'quit') or has_sword
  if chapter == 'kansas':
    if goal_castle:  # up the road through the gap in the castle wall (row 0, columns 6-8)
      if row > 5:
        return 0
      if col != 7 and row >= 1:
        return 2 if col > 7 else 3
      return 0
    if row != 5:  # to the cavern: along the road of row 5 to the east edge
      return 0 if row > 5 else 1
    return 3
  if chapter == 'cavern':
    if not has_sword:  # the sword lies at (3, 8); row 4 is open from the entrance
      if col < 8:
        return 1 if row < 4 else (0 if row > 4 else 3)
      if col > 8:
        return 2
      return 0 if row > 3 else 1
    if row != 4:
      return 1 if row < 4 else 0
    return 2
  # castle: towards the dragonduck, or around
  return int(rng.choice([0, 0, 2, 3, 1]))


class OracleOrdeal(object):
  """The example's Story over oracle engines, one environment."""

  def __init__(self):
    from pycolab_amd import cropping
    from pycolab_amd.compiler import GameTemplate
    self.templates = {k: GameTemplate.load(os.path.join(ROOT, 'tests', 'golden', 'templates', 'ordeal_%s.npz' % k)) for k in KEYS}
    self.cropper_of = {'kansas': lambda: cropping.ScrollingCropper(rows=8, cols=15, to_track='P', scroll_margins=(2, 3))}  # ordeal.py:103-105
    self.game_over, self.this_chapter, self.eng, self.crop = False, None, None, None
    self.words = None  # the Plot entries that travel (storytelling.py:449-450)

  def _start(self, chapter, prior):
    if self.eng is not None:
      self.eng.close()
    self.this_chapter = chapter
    self.eng = binding.OracleEngine(self.templates[chapter], 1)
    words = np.zeros((N.PLOT_WORDS, 1), np.int32)
    words[N.PLOT_OD_LAST_POSITION] = words[N.PLOT_OD_PRIOR_CHAPTER] = -1
    if self.words is not None:
      words[:] = self.words
    words[N.PLOT_OD_PRIOR_CHAPTER] = -1 if prior is None else KEYS.index(prior)  # new_plot.prior_chapter (:453)
    self.eng.set_plot_words(words)
    self.crop = binding.OracleCropper(self.eng, self.cropper_of[chapter]()) if chapter in self.cropper_of else None
    self.eng.reset()  # its_showtime()

  def _now(self):
    e = self.eng
    if self.crop is not None:
      board = np.array(self.crop.crop()[0])[0, 0].copy()
    else:
      board = np.array(e.planes)[0, 0].copy()
    self.words = e.plot_words()
    return board, (float(e.reward[0]) if e.reward_set[0] else None), float(e.discount[0]), bool(e.done[0])

  @property
  def has_sword(self):
    return bool(self.words[N.PLOT_OD_HAS_SWORD, 0])

  @property
  def last_position(self):
    w = int(self.words[N.PLOT_OD_LAST_POSITION, 0])
    return (np.int16(w & 0xFFFF).item(), np.int16((w >> 16) & 0xFFFF).item())

  def _chain(self, board, reward, discount):  # storytelling.py:421-467 _start_next_game
    while True:
      n = int(self.eng.next_chapter()[0])
      if n in (N.CHAPTER_UNSET, N.CHAPTER_NONE):  # (a dict of chapters: a new Plot's next_chapter is None)
        self.game_over = True
        return board, reward, discount
      self._start(KEYS[n], self.this_chapter)
      board, more, discount, over = self._now()
      if more is not None:
        reward = more if reward is None else reward + more
      if not over:
        return board, reward, discount

  def its_showtime(self):
    self._start('kansas', None)  # ordeal.py:110 first_chapter
    board, reward, discount, over = self._now()
    return self._chain(board, reward, discount) if over else (board, reward, discount)

  def play(self, action):
    self.eng.step(np.array([action], np.int32), auto_reset=False)
    board, reward, discount, over = self._now()
    return self._chain(board, reward, discount) if over else (board, reward, discount)

  def close(self):
    if self.eng is not None:
      self.eng.close()
      self.eng = None
