"""The oracle against the reference ITSELF on randomised levels (CPU).  tests/test_random_levels.py draws levels of
warehouse_manager, better_scrolly_maze and scrolly_maze at random and holds the HIP path to the oracle on them; here
the same level -- same random art, same cast, same z-order -- is built a second time with the reference's own
ascii_art and the example files' own classes (imported from /root/reference or from oracle/_ref's bytecode) and stepped
live next to the oracle on random tapes (quit, None and out-of-range actions mixed in), resets included.  So the
checker of the randomised GPU tests is itself checked against the reference on the very shapes it is used on, not only
on the recorded fixtures."""
import importlib
import sys
import warnings

import numpy as np
import pytest

from oracle import binding, ref_live
from pycolab_amd.compiler import GameTemplate
from tests import test_random_levels as levels

pytestmark = pytest.mark.skipif(ref_live.reference_path() is None, reason='the reference is neither under /root/reference nor built under oracle/_ref')


def reference_kit():
  path = ref_live.reference_path()
  if path not in sys.path:
    sys.path.insert(0, path)
  warnings.filterwarnings('ignore', category=DeprecationWarning)
  art = importlib.import_module('pycolab.ascii_art')
  drapes = importlib.import_module('pycolab.prefab_parts.drapes')
  wm = importlib.import_module('pycolab.examples.warehouse_manager')
  bs = importlib.import_module('pycolab.examples.better_scrolly_maze')
  sm = importlib.import_module('pycolab.examples.scrolly_maze')
  hw = importlib.import_module('pycolab.examples.hello_world')
  em = importlib.import_module('pycolab.examples.extraterrestrial_marauders')

  # (the random levels start their patrollers in a direction that depends on the character: test_random_levels.py)
  class BetterPatroller(bs.PatrollerSprite):

    def __init__(self, corner, position, character):
      super(BetterPatroller, self).__init__(corner, position, character)
      self._moving_east = bool(ord(character) % 2)

  class Guard(sm.PatrollerSprite):

    def __init__(self, corner, position, character, virtual_position):
      super(Guard, self).__init__(corner, position, character, virtual_position)
      self._moving_east = bool(ord(character) % 2)

  return levels.Kit(ascii_art=art, Box=wm.BoxSprite, Judge=wm.JudgeDrape, Pusher=wm.PlayerSprite, Walker=bs.PlayerSprite,
                    Patroller=BetterPatroller, Cash=bs.CashDrape, Explorer=sm.PlayerSprite, Guard=Guard, Maze=sm.MazeDrape,
                    Coins=sm.CashDrape, Scrolly=drapes.Scrolly, Rolling=hw.RollingDrape, Sliding=hw.SlidingSprite,
                    EMPlayer=em.PlayerSprite, EMUpBolt=em.UpwardLaserBoltSprite, EMDownBolt=em.DownwardLaserBoltSprite,
                    EMMarauders=em.MarauderDrape, EMBunkers=em.BunkerDrape)


class _Unoccluded(object):
  """An ascii_art module whose games are built with occlusion_in_layers=False (engine.py:98; rendering.py:187-301)."""

  def __init__(self, module):
    import functools
    self.Partial = module.Partial
    self.ascii_art_to_game = functools.partial(module.ascii_art_to_game, occlusion_in_layers=False)


@pytest.mark.parametrize('occlusion', [True, False], ids=['occluded', 'unoccluded'])
@pytest.mark.parametrize('maker', [levels.random_warehouse, levels.random_better_scrolly, levels.random_scrolly, levels.random_hello,
                                   levels.random_marauders], ids=lambda m: m.__name__)
@pytest.mark.parametrize('seed', range(10))
def test_oracle_matches_the_live_reference_on_a_random_level(maker, seed, occlusion, monkeypatch):
  kit, ours = reference_kit(), levels.OURS
  if not occlusion:  # the layers are then the things' raw masks: compared plane by plane below
    kit = levels.Kit(**dict(kit.__dict__, ascii_art=_Unoccluded(kit.ascii_art)))
    ours = levels.Kit(**dict(ours.__dict__, ascii_art=_Unoccluded(ours.ascii_art)))
  template = GameTemplate.from_engine(maker(np.random.RandomState(7000 + seed), ours))
  make = lambda: maker(np.random.RandomState(7000 + seed), kit)
  n_actions, E, T = int(template.n_actions), 8, 160
  # the marauders' return fire: np.random.choice (extraterrestrial_marauders.py:253) as the counter-based draw the oracle
  # and the kernels implement (oracle/ref_live.py _Choice; the template carries the seed)
  template.param[0], template.param[1] = 0xFACE + seed, 0
  choice = ref_live._Choice(0xFACE + seed, binding.action_hash)
  monkeypatch.setattr(np.random, 'choice', choice)
  # uniform ordinary actions with the quit action (= n_actions in all three games), None and garbage mixed in
  rng = np.random.RandomState(8000 + seed)
  tape = rng.randint(0, n_actions, size=(T, E)).astype(np.int32)
  u = rng.rand(T, E)
  tape[u < 0.03] = n_actions
  tape[(u >= 0.03) & (u < 0.05)] = -1
  tape[(u >= 0.05) & (u < 0.07)] = rng.randint(n_actions + 1, 40)
  orc = binding.OracleEngine(template, E)
  orc.reset()
  now = lambda: (np.array(orc.planes).copy(), np.array(orc.reward), np.array(orc.reward_set), np.array(orc.discount), np.array(orc.done))
  frames = [now()]
  for t in range(T):
    orc.step(tape[t], auto_reset=True)
    assert not np.array(orc.error).any()
    frames.append(now())
  chars = [chr(c) for c in template.chars]
  ended = 0
  for e in range(E):
    choice.env = e
    game = make()
    obs, r, d = game.its_showtime()
    assert sorted(obs.layers) == sorted(chars)
    for t in range(T + 1):
      if t:
        if game.game_over:  # (a finished environment is rebuilt at the next step; that step's action is not used)
          game = make()
          obs, r, d = game.its_showtime()
          ended += 1
        else:
          a = int(tape[t - 1, e])
          obs, r, d = game.play(None if a < 0 else a)
      planes, reward, reward_set, discount, done = frames[t]
      where = '%s seed %d: env %d frame %d' % (maker.__name__, seed, e, t)
      assert np.array_equal(obs.board, planes[e, 0]), where
      # occluded: board == c (rendering.py:177-179); unoccluded: the raw masks
      assert np.array_equal(np.stack([obs.layers[c] for c in chars]), planes[e, 1:] != 0), where + ': layers'
      assert (r is None) == (not reward_set[e]) and (r or 0) == reward[e], where
      assert d == discount[e] and game.game_over == bool(done[e]), where
  assert ended > 0  # (every level sees restarts)


@pytest.mark.parametrize('maker,track', [(levels.random_warehouse, 'P'), (levels.random_warehouse, 'XP'), (levels.random_better_scrolly, 'bP'),
                                         (levels.random_better_scrolly, '@b'), (levels.random_scrolly, 'P'), (levels.random_scrolly, '@aP')],
                         ids=lambda x: getattr(x, '__name__', x))
@pytest.mark.parametrize('seed', range(6))
def test_oracle_croppers_match_the_live_reference_croppers_on_a_random_level(maker, track, seed):
  """Random windows (padded and not, larger than the board, off the board, every margin / offset / saccade setting,
  following sprites and drapes) of the reference's own croppers next to the oracle's, every step, resets included."""
  from pycolab_amd import cropping
  kit = reference_kit()
  ref_cropping = importlib.import_module('pycolab.cropping')
  template = GameTemplate.from_engine(maker(np.random.RandomState(7100 + seed)))
  make = lambda: maker(np.random.RandomState(7100 + seed), kit)
  have = {chr(sp['ch']) for sp in template.sprites} | {chr(d['ch']) for d in template.drapes}
  track = ''.join(c for c in track if c in have) or chr(template.sprites[0]['ch'])
  ours = levels._random_croppers(np.random.RandomState(7200 + seed), template, track, cropping)()
  n_actions, E, T = int(template.n_actions), 4, 100
  rng = np.random.RandomState(8100 + seed)
  tape = rng.randint(0, n_actions, size=(T, E)).astype(np.int32)
  tape[rng.rand(T, E) < 0.03] = n_actions
  orc = binding.OracleEngine(template, E)
  orc.reset()
  crops = [binding.OracleCropper(orc, c) for c in ours]
  windows = []
  for t in range(T + 1):
    if t:
      orc.step(tape[t - 1], auto_reset=True)
    per = [c.crop() for c in crops]
    assert not any(err.any() for _, err in per)
    windows.append([w for w, _ in per])
  chars = [chr(c) for c in template.chars]
  moved = 0
  for e in range(E):
    theirs = levels._random_croppers(np.random.RandomState(7200 + seed), template, track, ref_cropping)()
    game = make()
    for cr in theirs:
      cr.set_engine(game)
    obs = game.its_showtime()[0]
    for t in range(T + 1):
      if t:
        if game.game_over:
          game = make()
          for cr in theirs:
            cr.set_engine(game)
          obs = game.its_showtime()[0]
        else:
          obs = game.play(int(tape[t - 1, e]))[0]
      for i, cr in enumerate(theirs):
        out = cr.crop(obs)
        where = '%s seed %d: env %d frame %d cropper %d (%s)' % (maker.__name__, seed, e, t, i, type(cr).__name__)
        np.testing.assert_array_equal(out.board, windows[t][i][e, 0], err_msg=where)
        for k, c in enumerate(chars):
          np.testing.assert_array_equal(out.layers[c], windows[t][i][e, 1 + k] != 0, err_msg=where + ' layer ' + c)
        if t and not np.array_equal(windows[t][i][e, 0], windows[t - 1][i][e, 0]):
          moved += 1
  assert moved > 0


# ---- levels WITHOUT walls around them: numpy's index -1, IndexError past the last row / column, things off the board ------
# (the makers live in tests/test_random_levels.py, where the GPU suite steps the same levels through the kernels)
random_open_warehouse, random_open_better_scrolly, random_open_scrolly_maze = (
    levels.random_open_warehouse, levels.random_open_better_scrolly, levels.random_open_scrolly_maze)
random_open_warehouse_scenery = levels.random_open_warehouse_scenery


@pytest.mark.parametrize('maker', [random_open_warehouse, random_open_better_scrolly, random_open_scrolly_maze, random_open_warehouse_scenery], ids=lambda m: m.__name__)
@pytest.mark.parametrize('seed', range(12))
def test_oracle_matches_the_live_reference_on_a_random_unwalled_level(maker, seed):
  """Where the reference raises (the IndexError of `layers[..][row + 1, col]` past the last row, warehouse_manager.py:
  219-226, better_scrolly_maze.py:291-294), the oracle's error bit comes up in that frame; until then every board is the
  reference's -- pushes "through" index -1, patrollers turning on walls they see at [row, -1], boxes and players off the
  board at position (0, 0) included (tests/test_raise_parity.py check_walkers; the committed fixtures of this kind are
  warehouse_open_A / _B and better_scrolly_custom_D)."""
  from tests.test_raise_parity import check_walkers
  kit = reference_kit()
  template = GameTemplate.from_engine(maker(np.random.RandomState(7300 + seed), levels.OURS))
  make = lambda: maker(np.random.RandomState(7300 + seed), kit)
  n_actions, E, T = int(template.n_actions), 12, 120
  rng = np.random.RandomState(8300 + seed)
  tape = rng.randint(0, n_actions, size=(T, E)).astype(np.int32)
  tape[rng.rand(T, E) < 0.01] = n_actions  # (the quit action, rarely: episodes should last)
  boards, raise_frame, raise_bit = None, np.full(E, -1, np.int32), np.zeros(E, np.uint8)
  off_board = 0
  for e in range(E):
    game = make()
    obs = game.its_showtime()[0]
    if boards is None:
      boards = np.zeros((T + 1, E) + obs.board.shape, np.uint8)
    boards[0, e] = obs.board
    for t in range(T):
      try:
        if game.game_over:
          game = make()
          obs = game.its_showtime()[0]
        else:
          obs = game.play(int(tape[t, e]))[0]
      except IndexError:
        raise_frame[e], raise_bit[e] = t + 1, 1  # pcx_device.h ERR_INDEX
        break
      boards[t + 1, e] = obs.board
      off_board += sum(1 for th in game.things.values() if hasattr(th, 'virtual_position') and not th.visible)
  orc = binding.OracleEngine(template, E)
  orc.reset()

  def frames():
    for f in range(T + 1):
      if f:
        orc.step(tape[f - 1], auto_reset=True)
      yield f, np.array(orc.planes)[:, 0], np.array(orc.error)
  check_walkers(dict(boards=boards, raise_frame=raise_frame, raise_bit=raise_bit), frames())
  UNWALLED_STATS.append((maker.__name__, off_board, int((raise_frame >= 0).sum()), E))


UNWALLED_STATS = []


def test_the_random_unwalled_levels_see_things_off_the_board_and_raises():
  if not UNWALLED_STATS:
    pytest.skip('runs after test_oracle_matches_the_live_reference_on_a_random_unwalled_level in the same process')
  for name in ('random_open_warehouse', 'random_open_better_scrolly', 'random_open_scrolly_maze', 'random_open_warehouse_scenery'):
    rows = [s for s in UNWALLED_STATS if s[0] == name]
    if rows:
      assert sum(s[1] for s in rows) > 0 and 0 < sum(s[2] for s in rows) < sum(s[3] for s in rows), (name, rows)


@pytest.mark.parametrize('maker', [levels.random_warehouse, levels.random_better_scrolly, levels.random_marauders], ids=lambda m: m.__name__)
@pytest.mark.parametrize('seed', range(6))
def test_numpy_oracle_postprocessors_match_the_live_reference_on_a_random_level(maker, seed, monkeypatch):
  """oracle/postprocess.py (what the GPU suite checks the device post-processors and the fused epilogues against) next to
  the reference's own ObservationToArray / ObservationToFeatureArray / ObservationCharacterRepainter (rendering.py:304-661)
  on the observations of a random level: random value tables (scalars and vectors, several dtypes), random layer lists
  with characters the game does not have, every axis order, random repaint tables -- and the RuntimeError for a
  character without a value."""
  from oracle import postprocess as opost
  kit = reference_kit()
  rendering = importlib.import_module('pycolab.rendering')
  monkeypatch.setattr(np.random, 'choice', ref_live._Choice(0xBEE + seed, binding.action_hash))
  rng = np.random.RandomState(7600 + seed)
  game = maker(np.random.RandomState(7500 + seed), kit)
  obs = game.its_showtime()[0]
  chars = sorted(obs.layers)
  n_actions = 4
  for step in range(30):
    depth = int(rng.choice([0, 1, 3]))
    dtype = [np.float32, np.uint8, np.int32, np.float64][int(rng.randint(4))]
    values = {c: (rng.randint(0, 200, size=depth).tolist() if depth else float(rng.randint(0, 200))) for c in chars}
    permute = None if depth == 0 or rng.rand() < 0.4 else [int(x) for x in rng.permutation(3)]
    want = rendering.ObservationToArray(values, dtype=dtype, permute=permute)(obs)
    got = opost.to_array(obs.board, values, dtype, permute)
    assert got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want), ('to_array', step)
    missing = dict(values)
    gone = chr(int(rng.choice(np.unique(obs.board))))
    del missing[gone]
    with pytest.raises(RuntimeError):
      rendering.ObservationToArray(missing, dtype=dtype)(obs)
    with pytest.raises(RuntimeError):
      opost.to_array(obs.board, missing, dtype)
    layers = [c for c in chars if rng.rand() < 0.7] + ['~', 'q'][:int(rng.randint(3))]
    rng.shuffle(layers)
    if layers:
      permute = None if rng.rand() < 0.4 else [int(x) for x in rng.permutation(3)]
      want = rendering.ObservationToFeatureArray(layers, permute=permute)(obs)
      got = opost.feature_array(obs.layers, layers, obs.board.shape, permute)
      assert got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want), ('features', step)
    mapping = {c: chars[int(rng.randint(len(chars)))] if rng.rand() < 0.7 else '%' for c in chars if rng.rand() < 0.4}
    want = rendering.ObservationCharacterRepainter(mapping)(obs)
    board, got_layers = opost.repaint(obs.board, chars, mapping)
    assert np.array_equal(board, want.board) and sorted(got_layers) == sorted(want.layers), ('repaint', step)
    for c in want.layers:
      assert np.array_equal(got_layers[c], want.layers[c]), ('repaint layer', c, step)
    if game.game_over:
      break
    obs = game.play(int(rng.randint(n_actions)))[0]
