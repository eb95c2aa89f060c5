"""The oracle against the trace of the reference's own `storytelling.Story` (oracle/gen_story_golden.py), on the CPU: one
oracle engine per chapter game, chained the way storytelling.py:395-467 chains them -- the finished game's
`the_plot.next_chapter` (what the oracle's entities assigned, `ox_plot.next_chapter`, plot.py:299-324; unassigned: the
next chapter of the list) names the next game, rewards add up across the change of games, observation and discount are
the new game's.  (The GPU suite checks the product's `storytelling.Story` against the same traces and the device's
next_chapter against the oracle's; this is the missing side of the triangle: oracle <-> reference.)"""
import numpy as np
import pytest

from oracle import binding, directive_scenarios as ds
from pycolab_amd import _native as N
from tests import helpers

STORIES = {'story_three_chapters': ds.STORY, 'story_entity_chapters': ds.STORY_JUMPS}


def templates(specs):
  from pycolab_amd import ascii_art
  from pycolab_amd.compiler import GameTemplate
  from pycolab_amd.prefab_parts import tabled
  return [GameTemplate.from_engine(ds.build_twin(spec, ascii_art, tabled)) for spec in specs]


class OracleStory(object):
  """storytelling.Story (auto-advancing list of chapters) over one-environment oracle engines."""

  def __init__(self, chapter_templates):
    self.templates, self.game_over, self.this_chapter, self.eng = chapter_templates, False, None, None

  def _start(self, chapter):
    self.this_chapter = chapter
    self.eng = binding.OracleEngine(self.templates[chapter], 1)
    self.eng.reset()  # its_showtime()

  def _now(self):
    e = self.eng
    return (np.array(e.planes)[0, 0].copy(), int(e.reward[0]) if e.reward_set[0] else None, float(e.discount[0]), bool(e.done[0]))

  def _next_chapter(self):
    n = int(self.eng.next_chapter()[0])
    if n == N.CHAPTER_UNSET:  # storytelling.py:452-454: auto-advance
      n = self.this_chapter + 1
      return n if n < len(self.templates) else None
    return None if n == N.CHAPTER_NONE else n

  def _chain(self, board, reward, discount):  # storytelling.py:421-467 _start_next_game
    while True:
      nxt = self._next_chapter()
      if nxt is None:
        self.game_over = True
        return board, reward, discount
      self._start(nxt)
      board, more, discount, over = self._now()
      if more is not None:
        reward = more if reward is None else reward + more
      if not over:
        return board, reward, discount

  def its_showtime(self):
    self._start(0)
    board, reward, discount, over = self._now()
    return self._chain(board, reward, discount) if over else (board, reward, discount)

  def play(self, action):
    self.eng.step(np.array([action], np.int32), auto_reset=False)
    board, reward, discount, over = self._now()
    return self._chain(board, reward, discount) if over else (board, reward, discount)


@pytest.mark.parametrize('which', sorted(STORIES))
def test_oracle_story_matches_reference_story(which):
  tr = helpers.load_trace_raw(which)
  ts = templates(STORIES[which])
  T, E = tr['actions'].shape
  jumps = 0
  for e in range(E):
    story = OracleStory(ts)

    def check(out, row):
      board, r, d = out
      where = '%s: env %d row %d' % (which, e, row)
      np.testing.assert_array_equal(board, tr['boards'][row, e], err_msg=where)
      assert (r is None) == (not tr['reward_set'][row, e]) and (r or 0) == tr['reward'][row, e], where
      assert d == tr['discount'][row, e] and story.game_over == bool(tr['done'][row, e]), where
      if not story.game_over:
        assert story.this_chapter == tr['chapter'][row, e], where
    check(story.its_showtime(), 0)
    for t in range(T):
      if story.game_over:
        story = OracleStory(ts)
        check(story.its_showtime(), t + 1)
        continue
      before = story.this_chapter
      check(story.play(int(tr['actions'][t, e])), t + 1)
      jumps += (not story.game_over) and story.this_chapter not in (before, before + 1)
  assert (jumps > 0) == (which == 'story_entity_chapters')  # (entities sent the story somewhere else than "the next one")


@pytest.mark.parametrize('which', sorted(STORIES))
@pytest.mark.parametrize('seed', range(6))
def test_oracle_story_matches_the_live_reference_story_on_fresh_tapes(which, seed):
  """The same chain next to the reference's own `storytelling.Story` stepped live (from /root/reference or oracle/_ref)
  on tapes the fixture does not hold: 6 x 12 more stories of each kind, 120 steps each, restarted when they end."""
  import collections
  import collections.abc
  import importlib
  import sys
  import warnings
  from oracle import ref_live
  path = ref_live.reference_path()
  if path is None:
    pytest.skip('the reference is neither under /root/reference nor built under oracle/_ref')
  if path not in sys.path:
    sys.path.insert(0, path)
  warnings.filterwarnings('ignore', category=DeprecationWarning)
  for name in ('Mapping', 'Sequence'):  # storytelling.py uses collections.Mapping, gone since Python 3.10 (oracle/gen_story_golden.py)
    if not hasattr(collections, name):
      setattr(collections, name, getattr(collections.abc, name))
  ref_art = importlib.import_module('pycolab.ascii_art')
  ref_story = importlib.import_module('pycolab.storytelling')
  tt = importlib.import_module('pycolab.tests.test_things')
  specs = STORIES[which]
  ts = templates(specs)
  make_story = lambda: ref_story.Story([lambda spec=spec: ds.build_reference(spec, ref_art, tt) for spec in specs])
  T, chapters_seen = 120, set()
  for e in range(12):
    tape = ds.story_tape(np.random.RandomState(12000 + 100 * seed + e), T)
    ours, theirs = OracleStory(ts), make_story()

    def check(out, ref_out, row):
      (board, r, d), (obs, rr, rd) = out, ref_out
      where = '%s seed %d: story %d row %d' % (which, seed, e, row)
      np.testing.assert_array_equal(board, obs.board, err_msg=where)
      assert r == rr and d == rd and ours.game_over == theirs.game_over, where
      if not ours.game_over:
        assert ours.this_chapter == theirs.the_plot.this_chapter, where
        chapters_seen.add(ours.this_chapter)
    check(ours.its_showtime(), theirs.its_showtime(), 0)
    for t in range(T):
      if theirs.game_over:
        ours, theirs = OracleStory(ts), make_story()
        check(ours.its_showtime(), theirs.its_showtime(), t + 1)
        continue
      a = int(tape[t])
      spec = specs[theirs.the_plot.this_chapter]
      ds.inject(spec, theirs.current_game, a, tt)
      check(ours.play(a), theirs.play(ds.reference_action(spec, a)), t + 1)
  assert chapters_seen == {0, 1, 2}
