"""examples/ordeal.py on the GPU -- SURVEY.md section 8 f-4's cited game (round 6): a `storytelling.Story` of three games
whose entities add float rewards (ordeal.py:123, 187-190), keep 'has_sword' / 'last_position' in the Plot, read
the_plot.this_chapter / prior_chapter / frame, assign the_plot.next_chapter and change the z-order (:113-269).

The three chapter templates are the UNCHANGED example file compiled by pycolab_amd.compiler (oracle/gen_templates.py;
tests/test_host_api.py compiles the file again where the reference is present); the Story around them is assembled the way
ordeal.py:82-110 `make_game()` does.  Checked against the trace recorded from the reference (oracle/gen_ordeal_golden.py)
at batch 1 -- reference types and the Plot dict -- and at batch 16 with every environment in its own chapter; the step
kernel (both builds of pcx_generic_step) against the oracle on hashed actions with staged plot words."""
import numpy as np
import pytest

from oracle import binding, ordeal_story
from pycolab_amd import _native as N
from tests import helpers

pytestmark = pytest.mark.gpu
KEYS = ordeal_story.KEYS


def make_ordeal(batch, auto_reset=False):
  from pycolab_amd import cropping, storytelling
  from pycolab_amd.engine import Engine
  chapters = {k: (lambda k=k: Engine.from_template(helpers.load_template('ordeal_' + k), batch=batch)) for k in KEYS}
  crop_kansas = cropping.ScrollingCropper(rows=8, cols=15, to_track='P', scroll_margins=(2, 3))  # ordeal.py:103-105
  return storytelling.Story(chapters=chapters, croppers=dict(castle=None, cavern=None, kansas=crop_kansas),
                            first_chapter='kansas', auto_reset=auto_reset)


@pytest.mark.parametrize('build', helpers.BUILDS)
def test_ordeal_batch1_matches_the_reference_trace(build, monkeypatch):
  helpers.force_generic(monkeypatch, build)
  tr = helpers.load_trace_raw('ordeal_story')
  T, E = tr['actions'].shape
  for e in (1, 2, 6) if build == 'specialised' else (0, 1, 2, 5, 6, 13):  # (6 and 13: the stories that fetch the sword first)
    story = make_ordeal(1)

    def check(out, row):
      obs, r, d = out
      where = 'env %d row %d' % (e, row)
      np.testing.assert_array_equal(obs.board, tr['boards'][row, e], err_msg=where)
      assert (r is None) == (not tr['reward_set'][row, e]) and (r or 0.0) == tr['reward'][row, e], where
      assert r is None or type(r) is float, where
      assert d == tr['discount'][row, e] and story.game_over == bool(tr['done'][row, e]), where
      if not story.game_over:
        plot = story.the_plot
        assert plot.this_chapter == KEYS[tr['chapter'][row, e]], where
        assert bool(plot.get('has_sword')) == bool(tr['has_sword'][row, e]), where
        assert tuple(plot['last_position']) == (tr['last_row'][row, e], tr['last_col'][row, e]), where
    check(story.its_showtime(), 0)
    for t in range(T):
      if story.game_over:
        story.close()
        story = make_ordeal(1)
        check(story.its_showtime(), t + 1)
        continue
      check(story.play(int(tr['actions'][t, e])), t + 1)
    story.close()


@pytest.mark.parametrize('build', helpers.BUILDS)
def test_ordeal_batched_every_environment_in_its_own_chapter(build, monkeypatch):
  helpers.force_generic(monkeypatch, build)
  tr = helpers.load_trace_raw('ordeal_story')
  T, E = tr['actions'].shape
  story = make_ordeal(E, auto_reset=True)
  chapters_together = 0

  def check(result, row):
    obs, reward, discount = result
    where = 'row %d' % row
    board = helpers.to_np(obs.board)
    np.testing.assert_array_equal(board, tr['boards'][row], err_msg=where)
    np.testing.assert_array_equal(story.reward_set, tr['reward_set'][row].astype(bool), err_msg=where)
    assert reward.dtype == np.float32
    np.testing.assert_array_equal(reward, tr['reward'][row], err_msg=where)
    np.testing.assert_array_equal(discount, tr['discount'][row], err_msg=where)
    np.testing.assert_array_equal(np.asarray(story.game_over), tr['done'][row].astype(bool), err_msg=where)
    assert story.this_chapter == [None if c < 0 else KEYS[c] for c in tr['chapter'][row]], where
    for key in set(story.this_chapter) - {None}:  # the Plot entries, where the environments are
      members = np.array([c == key for c in story.this_chapter])
      words = story.engine_of(key).plot_words()[:, members]
      np.testing.assert_array_equal(words[N.PLOT_OD_HAS_SWORD], tr['has_sword'][row][members], err_msg=where)
      np.testing.assert_array_equal(words[N.PLOT_OD_LAST_POSITION] & 0xFFFF, tr['last_row'][row][members], err_msg=where)
      np.testing.assert_array_equal(words[N.PLOT_OD_LAST_POSITION] >> 16, tr['last_col'][row][members], err_msg=where)
    return len(set(story.this_chapter) - {None})
  check(story.its_showtime(), 0)
  for t in range(T):
    chapters_together = max(chapters_together, check(story.play(tr['actions'][t]), t + 1))
  assert chapters_together == 3
  story.close()


@pytest.mark.parametrize('build', helpers.BUILDS)
@pytest.mark.parametrize('key', KEYS)
def test_ordeal_chapter_kernels_match_the_oracle_on_hashed_actions(key, build, monkeypatch):
  """One chapter's engine by itself, 512 environments x 96 hashed steps with auto-reset, every environment starting (and
  restarting) from its own staged plot words: planes, float reward bits, discount, game_over, plot words and the
  entities' next_chapter against the oracle."""
  from pycolab_amd.engine import Engine
  helpers.force_generic(monkeypatch, build)
  t = helpers.load_template('ordeal_' + key)
  B, T, seed = 512, 96, 0x0DEA1
  rng = np.random.RandomState(77)
  words = np.zeros((N.PLOT_WORDS, B), np.int32)
  words[N.PLOT_OD_HAS_SWORD] = rng.randint(0, 2, B)
  words[N.PLOT_OD_LAST_POSITION] = rng.randint(0, t.rows, B) | (rng.randint(0, t.cols, B) << 16)
  words[N.PLOT_OD_PRIOR_CHAPTER] = rng.randint(-1, 3, B)
  eng = Engine.from_template(t, batch=B, auto_reset=True)
  orc = binding.OracleEngine(t, B)
  eng.its_showtime()
  orc.reset()
  mask = rng.randint(0, 4, B) > 0  # a quarter of the environments keep a new Plot
  eng.set_plot_words(words, mask)
  orc.set_plot_words(words, mask)
  eng.reset()
  orc.reset()
  rewards = set()
  for step in range(T + 1):
    where = '%s step %d' % (key, step)
    np.testing.assert_array_equal(helpers.to_np(eng.planes_view()), orc.planes, err_msg=where)
    sc = eng._read_scalars()
    assert sc['reward'].dtype == np.float32
    np.testing.assert_array_equal(sc['reward'].view(np.int32), orc.reward.view(np.int32), err_msg=where)
    for k in ('reward_set', 'discount', 'done', 'frame', 'error'):
      np.testing.assert_array_equal(sc[k], getattr(orc, k), err_msg=where + ' ' + k)
    np.testing.assert_array_equal(eng.plot_words(), orc.plot_words(), err_msg=where)
    np.testing.assert_array_equal(eng.entities_next_chapter(), orc.next_chapter(), err_msg=where)
    rewards |= set(sc['reward'][sc['reward_set'] != 0].tolist())
    if step < T:
      eng.step_hashed(seed, step, 1)
      orc.step_hashed(seed, step, 1)
  assert N.lib().pcx_engine_kernel_name(eng._native).decode() == 'pcx_generic_step'
  if key == 'castle':
    assert rewards == {1.0, -1.0}
  eng.close()
  orc.close()


def test_ordeal_checkpoint_carries_the_plot_words_and_what_resets_start_from(monkeypatch):
  """`Engine.export_state()` of a chapter engine holds the plot words (state words) AND the staged words its auto-resets
  start from (`pcx_engine_set_plot_words`): a fresh engine that imports the checkpoint continues bit for bit, resets
  included."""
  from pycolab_amd.engine import Engine
  helpers.force_generic(monkeypatch, 'table-driven')
  t = helpers.load_template('ordeal_castle')
  B, seed = 256, 0xC4EC
  rng = np.random.RandomState(5)
  words = np.zeros((N.PLOT_WORDS, B), np.int32)
  words[N.PLOT_OD_HAS_SWORD] = rng.randint(0, 2, B)
  words[N.PLOT_OD_LAST_POSITION] = rng.randint(0, t.rows, B) | (rng.randint(6, 9, B) << 16)
  words[N.PLOT_OD_PRIOR_CHAPTER] = 2  # from Kansas: the player enters through the gate, in the column it left
  a = Engine.from_template(t, batch=B, auto_reset=True)
  a.its_showtime()
  a.set_plot_words(words)
  a.reset()
  a.step_hashed(seed, 0, 24)
  blob = a.export_state()
  b = Engine.from_template(t, batch=B, auto_reset=True)
  b.its_showtime()
  b.import_state(blob)
  np.testing.assert_array_equal(b.plot_words(), a.plot_words())
  resets = 0
  for step in range(24, 72):
    a.step_hashed(seed, step, 1)
    b.step_hashed(seed, step, 1)
    np.testing.assert_array_equal(helpers.to_np(b.planes_view()), helpers.to_np(a.planes_view()), err_msg='step %d' % step)
    np.testing.assert_array_equal(b.plot_words(), a.plot_words(), err_msg='step %d' % step)
    sa, sb = a._read_scalars(), b._read_scalars()
    for k in sa:
      np.testing.assert_array_equal(sb[k], sa[k], err_msg='step %d %s' % (step, k))
    fresh = sa['frame'] == 0
    resets += int(fresh.sum())
    # an environment that started over did so from ITS staged words: with the sword where it was staged, behind the gate's column
    np.testing.assert_array_equal(a.plot_words()[N.PLOT_OD_HAS_SWORD][fresh], words[N.PLOT_OD_HAS_SWORD][fresh])
    np.testing.assert_array_equal(a.plot_words()[N.PLOT_OD_LAST_POSITION][fresh] >> 16, words[N.PLOT_OD_LAST_POSITION][fresh] >> 16)
  assert resets > B // 4
  a.close()
  b.close()
