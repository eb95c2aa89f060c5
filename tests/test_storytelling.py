"""storytelling.Story over device engines against a trace of the reference's own
Story (oracle/gen_story_golden.py: three chapters built from the reference's
test entities, Plot directives ending them with default and custom discounts).
Batch 1 replays every environment of the trace with the reference's semantics;
batch 16 runs all of them at once, every environment in its own chapter."""
import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


def chapters(batch, which='story_three_chapters'):
  from oracle import directive_scenarios as ds
  from pycolab_amd import ascii_art
  from pycolab_amd.prefab_parts import tabled
  specs = ds.STORY if which == 'story_three_chapters' else ds.STORY_JUMPS
  return [lambda spec=spec: ds.build_twin(spec, ascii_art, tabled).configure(batch=batch) for spec in specs]


# story_entity_chapters: the games' entities assign the_plot.next_chapter themselves (plot.py:299-324,
# as examples/ordeal.py:177-235 does) -- recorded from the reference's own Story
STORIES = ['story_three_chapters', 'story_entity_chapters']


@pytest.mark.parametrize('which', STORIES)
def test_story_batch1_matches_reference_story(which):
  from pycolab_amd import storytelling
  tr = helpers.load_trace_raw(which)
  T, E = tr['actions'].shape
  for e in range(0, E, 3):
    story = storytelling.Story(chapters(1, which))
    obs, r, d = story.its_showtime()
    row = 0

    def check(obs, r, d, row):
      where = 'env %d row %d' % (e, row)
      np.testing.assert_array_equal(obs.board, tr['boards'][row, e], err_msg=where)
      assert (r is None) == (not tr['reward_set'][row, e]) and (r or 0) == tr['reward'][row, e], where
      assert d == tr['discount'][row, e] and story.game_over == bool(tr['done'][row, e]), where
      if not story.game_over:
        assert story.the_plot.this_chapter == tr['chapter'][row, e], where
    check(obs, r, d, 0)
    for t in range(T):
      row = t + 1
      if story.game_over:
        with pytest.raises(RuntimeError):
          story.play(0)
        story.close()
        story = storytelling.Story(chapters(1, which))
        check(*story.its_showtime(), row)
        continue
      check(*story.play(int(tr['actions'][t, e])), row)
    story.close()


@pytest.mark.parametrize('which', STORIES)
def test_story_batched_every_environment_in_its_own_chapter(which):
  from pycolab_amd import storytelling
  tr = helpers.load_trace_raw(which)
  T, E = tr['actions'].shape
  story = storytelling.Story(chapters(E, which), auto_reset=True)
  chars = [chr(c) for c in tr['chars']]

  def check(result, row):
    obs, reward, discount = result
    board = helpers.to_np(obs.board)
    np.testing.assert_array_equal(board, tr['boards'][row], err_msg='row %d' % row)
    np.testing.assert_array_equal(story.reward_set, tr['reward_set'][row].astype(bool), err_msg='row %d' % row)
    np.testing.assert_array_equal(reward, tr['reward'][row], err_msg='row %d' % row)
    np.testing.assert_array_equal(discount, tr['discount'][row], err_msg='row %d' % row)
    np.testing.assert_array_equal(np.asarray(story.game_over), tr['done'][row].astype(bool), err_msg='row %d' % row)
    want_ch = [None if c < 0 else int(c) for c in tr['chapter'][row]]
    assert story.this_chapter == want_ch, 'row %d' % row
    assert sorted(obs.layers) == sorted(chars)
    for ch in chars:   # layers over the union of the chapters' characters
      np.testing.assert_array_equal(helpers.to_np(obs.layers[ch]), (board == ord(ch)).astype(np.uint8))
  check(story.its_showtime(), 0)
  for t in range(T):
    check(story.play(tr['actions'][t]), t + 1)
  story.close()


def test_story_batched_with_one_cropper_shared_by_all_chapters():
  """The reference lets ONE cropper serve every game of a Story (storytelling.py:129-137).
  At batch > 1 every chapter has its own live engine, so the story clones the cropper per
  chapter: each environment's window must come from the chapter IT is in."""
  from pycolab_amd import cropping, storytelling
  tr = helpers.load_trace_raw('story_three_chapters')
  T, E = tr['actions'].shape
  shared = cropping.FixedCropper((-1, 3), 4, 5, pad_char='.')
  story = storytelling.Story(chapters(E), croppers=shared, auto_reset=True)
  assert len({id(c) for c in story._croppers.values()}) == 3
  assert story.rows == 4 and story.cols == 5

  def want(row):
    full = tr['boards'][row]
    out = np.full((E, 4, 5), ord('.'), np.uint8)
    out[:, 1:, :] = full[:, 0:3, 3:8]
    return out

  obs = story.its_showtime()[0]
  np.testing.assert_array_equal(helpers.to_np(obs.board), want(0))
  for t in range(T):
    obs = story.play(tr['actions'][t])[0]
    np.testing.assert_array_equal(helpers.to_np(obs.board), want(t + 1), err_msg='row %d' % (t + 1))
  things = story.things  # storytelling.py:344-377: stand-ins for the other chapters' characters
  assert set(things) == set('QRabDc')
  assert sum(storytelling.is_fictional(x) for x in things.values()) == 4
  story.close()


@pytest.mark.parametrize('chapter', [0, 1, 2])
def test_entities_next_chapter_matches_oracle(chapter):
  """The PCX_DIR_NEXT_CHAPTER directive on its own: what the entities of each chapter game leave in
  the_plot.next_chapter, HIP engine against the CPU oracle on the same random directive tapes."""
  from oracle import binding, directive_scenarios as ds
  from pycolab_amd import _native as N, ascii_art
  from pycolab_amd.compiler import GameTemplate
  from pycolab_amd.prefab_parts import tabled
  from tests.hip_adapter import HipAdapter
  t = GameTemplate.from_engine(ds.build_twin(ds.STORY_JUMPS[chapter], ascii_art, tabled))
  B, T = 200, 60
  rng = np.random.RandomState(50 + chapter)
  tape = np.stack([ds.story_tape(rng, T) for _ in range(B)], axis=1)
  hip, orc = HipAdapter(t, B), binding.OracleEngine(t, B)
  hip.reset(); orc.reset()
  seen = set()
  for step in range(T):
    auto = step % 7 != 3
    hip.step(tape[step], auto_reset=auto); orc.step(tape[step], auto_reset=auto)
    got, want = hip.eng.entities_next_chapter(), orc.next_chapter()
    np.testing.assert_array_equal(got, want, err_msg='step %d' % step)
    np.testing.assert_array_equal(hip.read('planes'), np.array(orc.planes), err_msg='step %d' % step)
    np.testing.assert_array_equal(hip.read('done'), np.array(orc.done), err_msg='step %d' % step)
    seen.update(int(v) for v in np.unique(got))
  assert N.CHAPTER_UNSET in seen and len(seen) >= 3, seen


def test_story_constructor_checks():
  """storytelling.py:493-553, 556-624."""
  from pycolab_amd import ascii_art, storytelling
  from pycolab_amd.prefab_parts import tabled
  with pytest.raises(ValueError):
    storytelling.Story([])
  with pytest.raises(ValueError):
    storytelling.Story(chapters(1), first_chapter=7)
  big = lambda: ascii_art.ascii_art_to_game(['....', '.a..'], '.', sprites=dict(a=tabled.StaticSprite))
  with pytest.raises(ValueError):   # observations of different shapes
    storytelling.Story(chapters(1) + [big])
  clash = lambda: ascii_art.ascii_art_to_game(['.........', '...Q.....', '.........'], '.', drapes=dict(Q=tabled.StaticDrape))
  with pytest.raises(ValueError):   # 'Q' is a Sprite in one game and a Drape in another
    storytelling.Story(chapters(1) + [clash])
  story = storytelling.Story({'x': chapters(1)[0], 'y': chapters(1)[2]}, first_chapter='x')
  story.its_showtime()
  story.the_plot.next_chapter = 'y'      # entities are device programs: the host names the next chapter
  obs, r, d = story.play(2)              # Q terminates chapter 'x'
  assert story.the_plot.this_chapter == 'y' and story.the_plot.prior_chapter == 'x' and not story.game_over
  assert bytes(obs.board[0]) == b'....Q....' and d == 1.0
  story.the_plot.next_chapter = 'nowhere'
  with pytest.raises(KeyError):
    story.play(2)
