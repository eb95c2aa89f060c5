"""examples/ordeal.py -- SURVEY.md section 8 f-4's cited game: a Story of three games whose entities add FLOAT rewards, keep
'has_sword' / 'last_position' in the Plot, read this_chapter / prior_chapter and name the next chapter -- on the CPU: the
oracle's three programs (oracle/pcx_oracle.c prog_od_*) chained like storytelling.py:391-470 (oracle/ordeal_story.py),
against the trace recorded from the reference (oracle/gen_ordeal_golden.py) and against the reference stepped live on
tapes the fixture does not hold."""
import collections
import collections.abc
import importlib
import sys
import warnings

import numpy as np
import pytest

from oracle import ordeal_story
from tests import helpers


def _check_row(story, out, tr, row, e):
  board, r, d = out
  where = 'env %d row %d' % (e, row)
  np.testing.assert_array_equal(board, tr['boards'][row, e], err_msg=where)
  assert (r is None) == (not tr['reward_set'][row, e]) and (r or 0.0) == tr['reward'][row, e], where
  assert r is None or isinstance(r, float), where  # ordeal.py:123, 187-190: the rewards are floats
  assert d == tr['discount'][row, e] and story.game_over == bool(tr['done'][row, e]), where
  if not story.game_over:
    assert ordeal_story.KEYS.index(story.this_chapter) == tr['chapter'][row, e], where
    assert story.has_sword == bool(tr['has_sword'][row, e]), where
    assert story.last_position == (tr['last_row'][row, e], tr['last_col'][row, e]), where


def test_oracle_ordeal_matches_the_reference_trace():
  tr = helpers.load_trace_raw('ordeal_story')
  T, E = tr['actions'].shape
  assert {(int(c), float(r)) for c, r, s in zip(tr['chapter'].ravel(), tr['reward'].ravel(), tr['reward_set'].ravel()) if s} >= {
      (1, 1.0), (-1, 1.0), (-1, -1.0)}  # the sword picked up; the dragonduck slain and the player eaten (both end the story)
  for e in range(E):
    story = ordeal_story.OracleOrdeal()
    _check_row(story, story.its_showtime(), tr, 0, e)
    for t in range(T):
      if story.game_over:
        story.close()
        story = ordeal_story.OracleOrdeal()
        _check_row(story, story.its_showtime(), tr, t + 1, e)
        continue
      _check_row(story, story.play(int(tr['actions'][t, e])), tr, t + 1, e)
    story.close()


@pytest.mark.parametrize('seed', range(4))
def test_oracle_ordeal_matches_the_live_reference_on_fresh_tapes(seed):
  from oracle import ref_live
  path = ref_live.reference_path()
  if path is None:
    pytest.skip('the reference is neither under /root/reference nor built under oracle/_ref')
  if path not in sys.path:
    sys.path.insert(0, path)
  warnings.filterwarnings('ignore', category=DeprecationWarning)
  for name in ('Mapping', 'Sequence'):  # storytelling.py uses collections.Mapping, gone since Python 3.10
    if not hasattr(collections, name):
      setattr(collections, name, getattr(collections.abc, name))
  ref = importlib.import_module('pycolab.examples.ordeal')
  T, seen = 260, collections.Counter()
  for e in range(6):
    rng = np.random.RandomState(31000 + 100 * seed + e)
    style = ordeal_story.style_of(e + seed)
    ours, theirs = ordeal_story.OracleOrdeal(), ref.make_game()

    def check(out, ref_out, row):
      (board, r, d), (obs, rr, rd) = out, ref_out
      where = 'seed %d story %d (%s) row %d' % (seed, e, style, row)
      np.testing.assert_array_equal(board, obs.board, err_msg=where)
      assert r == rr and (r is None) == (rr is None) and d == rd and ours.game_over == theirs.game_over, where
      if not ours.game_over:
        assert ours.this_chapter == theirs.the_plot.this_chapter, where
        assert ours.has_sword == bool(theirs.the_plot.get('has_sword')), where
        assert ours.last_position == tuple(theirs.the_plot['last_position']), where
      seen[(theirs.the_plot.this_chapter, rr)] += 1
    check(ours.its_showtime(), theirs.its_showtime(), 0)
    for t in range(T):
      if theirs.game_over:
        ours.close()
        ours, theirs = ordeal_story.OracleOrdeal(), ref.make_game()
        check(ours.its_showtime(), theirs.its_showtime(), t + 1)
        continue
      p = theirs.current_game.things['P'].position
      a = ordeal_story.tape_action(rng, style, theirs.the_plot.this_chapter, p.row, p.col, bool(theirs.the_plot.get('has_sword')), t)
      check(ours.play(a), theirs.play(a), t + 1)
    ours.close()
  assert {k[0] for k in seen} == {'castle', 'cavern', 'kansas'} and seen[('castle', -1.0)] > 0, seen
