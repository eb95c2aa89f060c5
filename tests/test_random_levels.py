"""Randomised levels of the shipped games, built where the reference's example
files are absent: the entity classes below opt in to the device programs with
`pcx_program` (what a user's own variant of a shipped class does), the art is
drawn at random, and the HIP path must agree with the oracle on every step.
Exercises the shape-generic / run-time-shape instances of the hand-written
kernels on shapes no fixture has (boards that are not whole dwords, one to ten
boxes, one to six walkers, windows of 12 to 900 cells)."""
import os

import numpy as np
import pytest

from tests import helpers

from oracle import binding
from pycolab_amd import _native as N
from pycolab_amd import ascii_art
from pycolab_amd import things
from pycolab_amd.compiler import GameTemplate
from pycolab_amd.prefab_parts import drapes as prefab_drapes
from pycolab_amd.prefab_parts import sprites as prefab_sprites
from tests.hip_adapter import HipAdapter


class OracleAdapter(binding.OracleEngine):

  def read(self, name):
    return np.array(getattr(self, name))


# ---- warehouse_manager (constructors as warehouse_manager.py:209-212, 241-243, 280-283) ----
class Box(prefab_sprites.MazeWalker):
  pcx_program = 'warehouse.box'

  def __init__(self, corner, position, character):
    super(Box, self).__init__(corner, position, character, set('#.0123456789PX') - set(character))


class Judge(things.Drape):
  pcx_program = 'warehouse.judge'

  def __init__(self, curtain, character):
    super(Judge, self).__init__(curtain, character)
    self._last_num_boxes_on_goals = 0


class Pusher(prefab_sprites.MazeWalker):
  pcx_program = 'warehouse.player'

  def __init__(self, corner, position, character):
    super(Pusher, self).__init__(corner, position, character, impassable='#.0123456789X')


class Kit(object):
  """What a random level is built WITH: this package's ascii_art and the opt-in classes below (the default), or the
  reference's own module and example classes (tests/test_reference_live_random_levels.py) -- same art, same casts."""

  def __init__(self, **parts):
    self.__dict__.update(parts)


def random_warehouse(rng, kit=None):
  kit = kit or OURS
  rows, cols = int(rng.randint(5, 14)), int(rng.randint(6, 31))
  art = np.full((rows, cols), '.', dtype='<U1')
  art[1:-1, 1:-1] = '#'
  art[2:-2, 2:-2] = ' '
  inner = [(r, c) for r in range(2, rows - 2) for c in range(2, cols - 2)]
  if len(inner) < 4:
    return random_warehouse(rng, kit)
  rng.shuffle(inner)
  n_boxes = int(rng.randint(1, min(10, len(inner) // 3) + 1))
  cells = iter(inner)
  boxes = list('1234567890'[:n_boxes])
  for ch in boxes:
    art[next(cells)] = ch
  art[next(cells)] = 'P'
  for _ in range(n_boxes):
    art[next(cells)] = '_'
  for cell in cells:
    if rng.rand() < 0.12:
      art[cell] = '#'
  sprites = {ch: kit.Box for ch in boxes}
  sprites['P'] = kit.Pusher
  return kit.ascii_art.ascii_art_to_game([''.join(r) for r in art], ' ', sprites, {'X': kit.Judge},
                                         update_schedule=[boxes, ['X'], ['P']])


# ---- better_scrolly_maze (better_scrolly_maze.py:250-320) ----
class Walker(prefab_sprites.MazeWalker):
  pcx_program = 'better_scrolly_maze.player'

  def __init__(self, corner, position, character):
    super(Walker, self).__init__(corner, position, character, impassable='#')


class Patroller(prefab_sprites.MazeWalker):
  pcx_program = 'better_scrolly_maze.patroller'

  def __init__(self, corner, position, character):
    super(Patroller, self).__init__(corner, position, character, impassable='#')
    self._moving_east = bool(ord(character) % 2)


class Cash(things.Drape):
  pcx_program = 'better_scrolly_maze.cash'


def random_better_scrolly(rng, kit=None):
  kit = kit or OURS
  rows, cols = int(rng.randint(6, 40)), int(rng.randint(8, 70))
  art = np.full((rows, cols), ' ', dtype='<U1')
  art[0, :] = art[-1, :] = art[:, 0] = art[:, -1] = '#'
  inner = rng.rand(rows - 2, cols - 2)
  art[1:-1, 1:-1][inner < 0.2] = '#'
  coins = (inner >= 0.2) & (inner < 0.2 + min(0.1, 200.0 / (rows * cols)))
  art[1:-1, 1:-1][coins] = '@'
  free = [(r, c) for r in range(1, rows - 1) for c in range(1, cols - 1) if art[r, c] == ' ']
  if len(free) < 4 or not coins.any():
    return random_better_scrolly(rng, kit)
  rng.shuffle(free)
  for ch, cell in zip('abcP', free):
    art[cell] = ch
  return kit.ascii_art.ascii_art_to_game(
      [''.join(r) for r in art], ' ',
      sprites={'P': kit.Walker, 'a': kit.Patroller, 'b': kit.Patroller, 'c': kit.Patroller}, drapes={'@': kit.Cash},
      update_schedule=['a', 'b', 'c', 'P', '@'], z_order='abc@P')


# ---- scrolly_maze (scrolly_maze.py:245-357; built as scrolly_maze.make_game does, :213-235) ----
class Explorer(prefab_sprites.MazeWalker):
  pcx_program = 'scrolly_maze.player'

  def __init__(self, corner, position, character, virtual_position):
    super(Explorer, self).__init__(corner, position, character, egocentric_scroller=True, impassable='#')
    self._teleport(virtual_position)


class Guard(prefab_sprites.MazeWalker):
  pcx_program = 'scrolly_maze.patroller'

  def __init__(self, corner, position, character, virtual_position):
    super(Guard, self).__init__(corner, position, character, '#')
    self._teleport(virtual_position)
    self._moving_east = bool(ord(character) % 2)


class Maze(prefab_drapes.Scrolly):
  pcx_program = 'scrolly_maze.maze'


class Coins(prefab_drapes.Scrolly):
  pcx_program = 'scrolly_maze.cash'


def random_scrolly(rng, kit=None, walled=True):
  kit = kit or OURS
  br, bc = int(rng.randint(4, 13)), int(rng.randint(6, 33))  # (the default scroll margins (2, 3) need at least 4 x 6)
  rows, cols = br + int(rng.randint(0, 25)), bc + int(rng.randint(0, 50))
  rows, cols = max(rows, 5), max(cols, 6)
  br, bc = min(br, rows), min(bc, cols)
  patrollers = 'abcde'[:int(rng.randint(0, 6))]
  sprites = patrollers + 'P'
  art = np.full((rows, cols), ' ', dtype='<U1')
  if walled:  # (tests/test_reference_live_random_levels.py also draws worlds without a wall around them)
    art[0, :] = art[-1, :] = art[:, 0] = art[:, -1] = '#'
  inner = rng.rand(rows - 2, cols - 2)
  art[1:-1, 1:-1][inner < 0.2] = '#'
  art[1:-1, 1:-1][(inner >= 0.2) & (inner < 0.27)] = '@'
  cr, cc = int(rng.randint(0, rows - br + 1)), int(rng.randint(0, cols - bc + 1))
  placed = set()
  for ch in sprites:
    for _ in range(2000):
      if ch == 'P':  # the egocentric player starts inside the window
        r, c = int(rng.randint(cr, cr + br)), int(rng.randint(cc, cc + bc))
      else:
        r, c = int(rng.randint(1, rows - 1)), int(rng.randint(1, cols - 1))
      if 0 < r < rows - 1 and 0 < c < cols - 1 and art[r, c] == ' ' and (r, c) != (cr, cc):
        art[r, c] = ch
        placed.add(ch)
        break
  if 'P' not in placed:
    return random_scrolly(rng, kit, walled)
  sprites = ''.join(ch for ch in sprites if ch in placed)
  beneath = art[cr, cc] if art[cr, cc] in '# ' else ' '
  art[cr, cc] = '+'
  stars = np.full((br, bc), ' ', dtype='<U1')
  stars[rng.rand(br, bc) < 0.1] = '.'
  maze, stars = [''.join(r) for r in art], [''.join(r) for r in stars]
  info = kit.Scrolly.PatternInfo(maze, stars, board_northwest_corner_mark='+', what_lies_beneath=str(beneath))
  parts = {ch: kit.ascii_art.Partial(kit.Explorer if ch == 'P' else kit.Guard, info.virtual_position(ch)) for ch in sprites}
  z = list(sprites.replace('P', '')) + ['@', '#', 'P']
  rng.shuffle(z)
  return kit.ascii_art.ascii_art_to_game(
      stars, what_lies_beneath=' ', sprites=parts,
      drapes={'#': kit.ascii_art.Partial(kit.Maze, **info.kwargs('#')), '@': kit.ascii_art.Partial(kit.Coins, **info.kwargs('@'))},
      update_schedule=[['#'], list(sprites), ['@']], z_order=''.join(z))


OURS = Kit(ascii_art=ascii_art, Box=Box, Judge=Judge, Pusher=Pusher, Walker=Walker, Patroller=Patroller, Cash=Cash,
           Explorer=Explorer, Guard=Guard, Maze=Maze, Coins=Coins, Scrolly=prefab_drapes.Scrolly)


# ---- levels WITHOUT walls around them: numpy's index -1, the IndexError past the last row / column, things off the
# board at position (0, 0).  tests/test_reference_live_random_levels.py holds the oracle to the live reference on these
# makers (CPU); test_random_unwalled_levels_match_oracle below holds the kernels to the oracle on them (GPU).
def random_open_warehouse(rng, kit=None, scenery=False):
  kit = kit or OURS
  rows, cols = int(rng.randint(4, 9)), int(rng.randint(5, 11))
  art = np.full((rows, cols), ' ', dtype='<U1')
  art[rng.rand(rows, cols) < 0.08] = '#'
  cells = [(r, c) for r in range(rows) for c in range(cols) if art[r, c] == ' ']
  rng.shuffle(cells)
  boxes = list('1234'[:int(rng.randint(1, 5))])
  # scenery: all four backdrop-only characters of the shipped levels (' ', '#', '.', '_') -- what pcx_warehouse_step's
  # run-time-shape instance takes; without it (three characters, or two) the level is pcx_generic_step's
  extra = ['.'] * int(rng.randint(1, 4)) + ['#'] if scenery else []
  for ch in boxes + ['P'] + ['_'] * (len(boxes) + 1) + extra:
    art[cells.pop()] = ch
  sprites = {ch: kit.Box for ch in boxes}
  sprites['P'] = kit.Pusher
  return kit.ascii_art.ascii_art_to_game([''.join(r) for r in art], ' ', sprites, {'X': kit.Judge}, update_schedule=[boxes, ['X'], ['P']])


def random_open_warehouse_scenery(rng, kit=None):
  return random_open_warehouse(rng, kit, scenery=True)


def random_open_better_scrolly(rng, kit=None):
  kit = kit or OURS
  rows, cols = int(rng.randint(5, 10)), int(rng.randint(7, 15))
  art = np.full((rows, cols), ' ', dtype='<U1')
  u = rng.rand(rows, cols)
  art[u < 0.12] = '#'
  art[(u >= 0.12) & (u < 0.2)] = '@'
  art[:, -1][rng.rand(rows) < 0.6] = '#'  # (most rows end on a wall: fewer patrollers run into the IndexError of `col + 1`)
  art[0, 0] = '@'                         # (where everything off the board "is")
  cells = [(r, c) for r in range(rows) for c in range(cols - 1) if art[r, c] == ' ']
  rng.shuffle(cells)
  for ch in 'abcP':
    art[cells.pop()] = ch
  return kit.ascii_art.ascii_art_to_game(
      [''.join(r) for r in art], ' ', sprites={'P': kit.Walker, 'a': kit.Patroller, 'b': kit.Patroller, 'c': kit.Patroller},
      drapes={'@': kit.Cash}, update_schedule=['a', 'b', 'c', 'P', '@'], z_order='abc@P')


def random_open_scrolly_maze(rng, kit=None):
  """random_scrolly without the wall around the world: patrollers reach the pattern's first and last column
  (`whole_pattern[row, col - 1]` is the LAST column there, `[row, col + 1]` an IndexError: scrolly_maze.py:295-299), the
  window scrolls up to the pattern's bare edge."""
  return random_scrolly(rng, kit, walled=False)


def _compare(t, kernel, batch, steps, seed):
  hip, orc = HipAdapter(t, batch), OracleAdapter(t, batch)
  hip.reset(); orc.reset()
  assert N.lib().pcx_engine_kernel_name(hip.eng._native).decode() == kernel
  t0 = 0
  while t0 < steps:
    n = 1 if t0 < 12 else 6
    hip.step_hashed(seed, t0, n); orc.step_hashed(seed, t0, n)
    t0 += n
    for name in ('planes', 'reward', 'reward_set', 'discount', 'done', 'frame', 'error'):
      np.testing.assert_array_equal(hip.read(name), orc.read(name), err_msg='%s after step %d (%dx%d)' % (name, t0, t.rows, t.cols))
  np.testing.assert_array_equal(hip.sprites(), orc.sprites())
  # quirky actions: None, the quit action, out-of-range values; episodes left finished every third step
  rng = np.random.RandomState(seed)
  n_act = int(t.n_actions)
  for step in range(24):
    a = rng.randint(0, n_act, size=batch).astype(np.int32)
    r = rng.rand(batch)
    a[r < 0.05] = -1
    a[(r >= 0.05) & (r < 0.08)] = n_act
    a[(r >= 0.08) & (r < 0.11)] = rng.randint(n_act + 1, 40)
    auto = step % 3 != 0
    hip.step(a, auto_reset=auto); orc.step(a, auto_reset=auto)
    for name in ('planes', 'reward', 'reward_set', 'discount', 'done', 'frame', 'error'):
      np.testing.assert_array_equal(hip.read(name), orc.read(name), err_msg='%s, quirky step %d (%dx%d)' % (name, step, t.rows, t.cols))
  hip.eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize('maker,kernel', [(random_warehouse, 'pcx_warehouse_step'), (random_better_scrolly, 'pcx_better_scrolly_step'),
                                          (random_scrolly, 'pcx_scrolly_maze_step')])
@pytest.mark.parametrize('seed', range(8))
def test_random_levels_match_oracle(maker, kernel, seed):
  rng = np.random.RandomState(1000 + seed)
  t = GameTemplate.from_engine(maker(rng))
  _compare(t, kernel, batch=int(rng.choice([70, 200, 1500])), steps=60, seed=0xF00D + seed)



@pytest.mark.gpu
@pytest.mark.parametrize('build', helpers.BUILDS)
@pytest.mark.parametrize('maker', [random_warehouse, random_better_scrolly])
@pytest.mark.parametrize('seed', range(4))
def test_random_levels_match_oracle_through_the_table_driven_kernel(maker, seed, build, monkeypatch):
  helpers.force_generic(monkeypatch, build)
  rng = np.random.RandomState(2000 + seed)
  t = GameTemplate.from_engine(maker(rng))
  _compare(t, 'pcx_generic_step', batch=int(rng.choice([70, 200])), steps=48, seed=0xBEAD + seed)


UNWALLED_KERNELS = {}  # maker name -> kernels seen (the last test of the family asserts the hand-written ones were among them)


UNWALLED_MAKERS = [random_open_warehouse, random_open_warehouse_scenery, random_open_better_scrolly, random_open_scrolly_maze]
# every level through the kernel the engine picks; through both builds of pcx_generic_step the first three of every maker
# (an unwalled scrolly_maze world raises for every environment at once or for none: level seed 7 is one that raises)
UNWALLED_CASES = [(m, s, 'default') for m in UNWALLED_MAKERS for s in ((0, 1, 2, 3, 4, 7) if m is random_open_scrolly_maze else range(6))] + \
                 [(m, s, r) for m in UNWALLED_MAKERS[:3] for s in range(3) for r in ('table-driven', 'specialised')]


@pytest.mark.gpu
@pytest.mark.parametrize('maker,seed,route', UNWALLED_CASES, ids=lambda x: getattr(x, '__name__', str(x)))
def test_random_unwalled_levels_match_oracle(maker, seed, route, monkeypatch):
  """The GPU twin of test_oracle_matches_the_live_reference_on_a_random_unwalled_level (same makers, same level seeds): the
  kernel the engine picks by itself (`default`: pcx_warehouse_step's run-time-shape instance for the warehouses with the
  usual four backdrop characters, pcx_better_scrolly_step, pcx_scrolly_maze_step, pcx_generic_step for the others) and both
  builds of pcx_generic_step, against the oracle: the error bit comes up in the same frame with the same kind -- the frame
  the reference raises IndexError at -- and until then every output is equal, pushes through index -1, patrollers that look
  around (0, 0) from outside the board and off-board boxes on the goal at (0, 0) included."""
  if route != 'default':  # (scrolly_maze has no table-driven program: pcx_scrolly_maze_step steps every level of it)
    helpers.force_generic(monkeypatch, route)
  t = GameTemplate.from_engine(maker(np.random.RandomState(7300 + seed)))
  B, T, n_actions = 64 * 3 + 9, 100, int(t.n_actions)
  hip, orc = HipAdapter(t, B), binding.OracleEngine(t, B)
  hip.reset(); orc.reset()
  kernel = N.lib().pcx_engine_kernel_name(hip.eng._native).decode()
  if route != 'default':
    assert kernel == 'pcx_generic_step'
  else:
    UNWALLED_KERNELS.setdefault(maker.__name__, set()).add(kernel)
  rng = np.random.RandomState(8300 + seed)
  before = np.zeros(B, np.uint8)  # error bits so far: once the reference has raised, what an environment shows is nobody's law
  raised_at = []
  for step in range(T):
    a = rng.randint(0, n_actions, size=B).astype(np.int32)
    a[rng.rand(B) < 0.01] = n_actions  # (the quit action, rarely: episodes should last)
    hip.step(a, auto_reset=True); orc.step(a, auto_reset=True)
    err_h, err_o = hip.read('error'), np.array(orc.error)
    fresh = before == 0
    np.testing.assert_array_equal(err_h[fresh], err_o[fresh], err_msg='%s: error bits after step %d' % (kernel, step + 1))
    raised_at += [step + 1] * int(((err_o != 0) & fresh).sum())
    ok = err_o == 0
    for name in ('planes', 'reward', 'reward_set', 'discount', 'done', 'frame'):
      np.testing.assert_array_equal(hip.read(name)[ok], np.array(getattr(orc, name))[ok], err_msg='%s: %s after step %d' % (kernel, name, step + 1))
    before |= err_o
  ok = before == 0
  np.testing.assert_array_equal(hip.sprites()[ok], orc.sprites()[ok])
  UNWALLED_STATS.append((maker.__name__, route, len(raised_at), B))
  hip.eng.close()


UNWALLED_STATS = []


@pytest.mark.gpu
def test_the_random_unwalled_levels_raise_and_reach_the_hand_written_kernels():
  if not UNWALLED_STATS:
    pytest.skip('runs after test_random_unwalled_levels_match_oracle in the same process')
  for name in sorted({s[0] for s in UNWALLED_STATS}):
    rows = [s for s in UNWALLED_STATS if s[0] == name]
    assert 0 < sum(s[2] for s in rows) < sum(s[3] for s in rows), (name, rows)  # some environments raise, most do not
  if os.environ.get('PCX_FORCE_GENERIC') == '1':
    return
  want = {'random_open_warehouse_scenery': 'pcx_warehouse_step', 'random_open_better_scrolly': 'pcx_better_scrolly_step',
          'random_open_scrolly_maze': 'pcx_scrolly_maze_step', 'random_open_warehouse': 'pcx_generic_step'}
  for name, kernels in UNWALLED_KERNELS.items():
    assert want[name] in kernels, (name, kernels)


def _random_croppers(rng, t, track, cropping=None):
  if cropping is None:  # (tests/test_reference_live_random_levels.py passes the reference's module)
    from pycolab_amd import cropping
  R, C = t.rows, t.cols
  pad = chr(t.chars[int(rng.randint(len(t.chars)))])
  out = []
  for _ in range(int(rng.randint(1, 5))):
    kind = rng.randint(3)
    if kind == 0:
      out.append(('F', (int(rng.randint(-3, R)), int(rng.randint(-3, C))), int(rng.randint(1, 9)), int(rng.randint(1, 12)), pad))
    else:
      rows, cols = int(rng.randint(3, 8)), int(rng.randint(3, 12))
      padded = kind == 1 or rows > R or cols > C
      margins = (int(rng.randint(1, (rows + 1) // 2)) if rows > 2 else 1, int(rng.randint(1, (cols + 1) // 2)) if cols > 2 else 1)
      if 2 * margins[0] >= rows or 2 * margins[1] >= cols:
        margins = (1, 1)
      out.append(('S', rows, cols, track, pad if padded else None, margins,
                  (int(rng.randint(-2, 3)), int(rng.randint(-2, 3))), bool(rng.randint(2))))
  def build():
    made = []
    for spec in out:
      if spec[0] == 'F':
        made.append(cropping.FixedCropper(spec[1], spec[2], spec[3], pad_char=spec[4]))
      else:
        made.append(cropping.ScrollingCropper(spec[1], spec[2], list(spec[3]), pad_char=spec[4], scroll_margins=spec[5],
                                              initial_offset=spec[6], saccade=spec[7]))
    return made
  return build


@pytest.mark.gpu
@pytest.mark.parametrize('maker,track,generic', [(random_warehouse, 'P', False), (random_better_scrolly, 'bP', False),
                                                 (random_warehouse, 'XP', True), (random_better_scrolly, '@b', True),
                                                 (random_warehouse, 'XP', 'specialised'), (random_better_scrolly, '@b', 'specialised')])
@pytest.mark.parametrize('seed', range(6))
def test_random_levels_fused_croppers_equal_stand_alone(maker, track, generic, seed, monkeypatch):
  """Random windows (padded and not, larger than the board, off the board, every
  margin / offset / saccade setting) on random levels: the step kernel's own
  croppers against the stand-alone cropper kernels, every step.  `generic`:
  through the table-driven kernel, whose croppers also follow drapes (the
  track lists there start with one)."""
  import torch
  from pycolab_amd import cropping
  from pycolab_amd.engine import Engine
  if generic:
    helpers.force_generic(monkeypatch, 'specialised' if generic == 'specialised' else 'table-driven')
  rng = np.random.RandomState(3000 + seed)
  t = GameTemplate.from_engine(maker(rng))
  have = {chr(sp['ch']) for sp in t.sprites} | {chr(d['ch']) for d in t.drapes}
  track = ''.join(c for c in track if c in have) or chr(t.sprites[0]['ch'])
  build = _random_croppers(rng, t, track)
  B = int(rng.choice([70, 333]))
  a = Engine.from_template(t, batch=B, auto_reset=True, seed=9)
  b = Engine.from_template(t, batch=B, auto_reset=True, seed=9)
  ca, cb = build(), build()
  for cr in ca:
    cr.set_engine(a)
  for cr in cb:
    cr.set_engine(b)
  cropping.fuse_croppers(a, ca)
  oa, ob = a.its_showtime()[0], b.its_showtime()[0]
  assert all(cr._fused for cr in ca)
  n_act = int(t.n_actions)
  for step in range(40):
    for i, (x, y) in enumerate(zip(ca, cb)):
      wx, wy = x.crop(oa), y.crop(ob)
      assert torch.equal(wx.board, wy.board), 'step %d cropper %d (%s)' % (step, i, type(x).__name__)
      for ch in wy.layers:
        assert torch.equal(wx.layers[ch], wy.layers[ch]), 'step %d cropper %d layer %r' % (step, i, ch)
    acts = rng.randint(0, n_act, size=B).astype(np.int32)
    oa, ob = a.play(acts)[0], b.play(acts)[0]
  for x, y in zip(ca, cb):  # the windows never left the observation without a pad character ... or both noticed
    ex = ey = None
    try:
      x.check_errors()
    except RuntimeError as e:
      ex = e
    try:
      y.check_errors()
    except RuntimeError as e:
      ey = e
    assert (ex is None) == (ey is None)
  a.close(); b.close()


# ---- hello_world (hello_world.py:72-123): boards other than the two compiled ones go to the table-driven kernel ----
class Rolling(things.Drape):
  pcx_program = 'hello_world.rolling'


class Sliding(things.Sprite):
  pcx_program = 'hello_world.sliding'
  _DX = ([-1, 1, -1, 1], [-1, 1, -1, 1], [1, -1, 1, -1], [1, -1, 1, -1])
  _DY = ([-1, 1, 1, -1], [1, -1, -1, 1], [1, -1, -1, 1], [-1, 1, 1, -1])

  def __init__(self, corner, position, character, direction_set):
    super(Sliding, self).__init__(corner, position, character)
    self._dx = self._DX[direction_set]
    self._dy = self._DY[direction_set]


def random_hello(rng, kit=None):
  kit = kit or OURS
  rows, cols = int(rng.randint(2, 24)), int(rng.randint(2, 60))
  art = np.full((rows, cols), ' ', dtype='<U1')
  art[rng.rand(rows, cols) < 0.2] = '@'
  art[rng.rand(rows, cols) < 0.05] = '#'
  cells = [(r, c) for r in range(rows) for c in range(cols)]
  rng.shuffle(cells)
  n = int(rng.randint(1, min(4, len(cells) - 1) + 1))
  names = '1234'[:n]
  for ch, cell in zip(names, cells):
    art[cell] = ch
  if not (art == '@').any():
    art[cells[-1]] = '@'
  z = list(names) + ['@']
  rng.shuffle(z)
  return kit.ascii_art.ascii_art_to_game(
      [''.join(r) for r in art], ' ',
      sprites={ch: kit.ascii_art.Partial(kit.Sliding, int(rng.randint(4))) for ch in names},
      drapes={'@': kit.Rolling}, z_order=''.join(z))


OURS.Rolling, OURS.Sliding = Rolling, Sliding


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(6))
def test_random_hello_world_boards_match_oracle(seed):
  rng = np.random.RandomState(4000 + seed)
  t = GameTemplate.from_engine(random_hello(rng))
  _compare(t, 'pcx_generic_step', batch=int(rng.choice([70, 200])), steps=48, seed=0xCAFE + seed)


# ---- extraterrestrial_marauders (extraterrestrial_marauders.py:104-256): the shipped 16x39 cast, other layouts ----
class EMPlayer(prefab_sprites.MazeWalker):
  pcx_program = 'marauders.player'

  def __init__(self, corner, position, character):
    super(EMPlayer, self).__init__(corner, position, character, impassable='', confined_to_board=True)


class EMUpBolt(prefab_sprites.MazeWalker):
  pcx_program = 'marauders.upward_bolt'

  def __init__(self, corner, position, character):
    super(EMUpBolt, self).__init__(corner, position, character, impassable='')
    self._teleport((-1, -1))


class EMDownBolt(prefab_sprites.MazeWalker):
  pcx_program = 'marauders.downward_bolt'

  def __init__(self, corner, position, character):
    super(EMDownBolt, self).__init__(corner, position, character, impassable='')
    self._teleport((-1, -1))


class EMMarauders(things.Drape):
  pcx_program = 'marauders.marauder'

  def __init__(self, curtain, character):
    super(EMMarauders, self).__init__(curtain, character)
    self._dx = -1


class EMBunkers(things.Drape):
  pcx_program = 'marauders.bunker'


def random_marauders(rng, kit=None):
  kit = kit or OURS
  rows, cols = 16, 39  # (the hand-written kernel's cast and board; the layout is free)
  art = np.full((rows, cols), ' ', dtype='<U1')
  top = int(rng.randint(0, 4))
  for r in range(top, top + int(rng.randint(1, 6))):
    for c in range(int(rng.randint(1, 6)), cols - int(rng.randint(1, 6))):
      if rng.rand() < 0.45:
        art[r, c] = 'X'
  for c0 in range(int(rng.randint(1, 5)), cols - 4, int(rng.randint(5, 11))):
    h, w = int(rng.randint(1, 4)), int(rng.randint(2, 5))
    art[10:10 + h, c0:c0 + w] = 'B'
  art[int(rng.choice([13, 14, 15])), int(rng.randint(0, cols))] = 'P'
  if not (art == 'X').any():
    art[1, 5] = 'X'
  sprites = dict([('P', kit.EMPlayer)] + [(c, kit.EMUpBolt) for c in 'abcd'] + [(c, kit.EMDownBolt) for c in 'yz'])
  return kit.ascii_art.ascii_art_to_game([''.join(r) for r in art], ' ', sprites, dict(X=kit.EMMarauders, B=kit.EMBunkers),
                                         update_schedule=['P', 'B', 'X'] + list('abcdyz'))


OURS.EMPlayer, OURS.EMUpBolt, OURS.EMDownBolt, OURS.EMMarauders, OURS.EMBunkers = EMPlayer, EMUpBolt, EMDownBolt, EMMarauders, EMBunkers


@pytest.mark.gpu
@pytest.mark.parametrize('generic', [False, True, 'specialised'])
@pytest.mark.parametrize('seed', range(5))
def test_random_marauders_layouts_match_oracle(seed, generic, monkeypatch):
  if generic:
    helpers.force_generic(monkeypatch, 'specialised' if generic == 'specialised' else 'table-driven')
  rng = np.random.RandomState(5000 + seed)
  t = GameTemplate.from_engine(random_marauders(rng))
  t.param[0] = 0xA11CE + seed  # seeds the marauders' return fire (np.random.choice, :253)
  _compare(t, 'pcx_generic_step' if generic else 'pcx_marauders_step', batch=int(rng.choice([70, 300])), steps=90, seed=0xD1CE + seed)


def test_random_levels_build_and_step_on_the_oracle():
  """CPU: the makers produce valid games (host mirror + template compiler) and the oracle steps them."""
  rng = np.random.RandomState(3)
  for maker in (random_warehouse, random_better_scrolly, random_scrolly, random_hello, random_marauders):
    for _ in range(4):
      t = GameTemplate.from_engine(maker(rng))
      orc = OracleAdapter(t, 8)
      orc.reset()
      orc.step_hashed(1, 0, 40)
      assert orc.read('planes').shape[1] == 1 + len(t.chars)
      assert not orc.read('error').any()


def test_nth_set_column_by_popcount_bisection():
  """pcx_generic.hip prog_em_downbolt picks the n-th set column of a 64-bit column mask (the marauders' return fire,
  extraterrestrial_marauders.py:246-248) with six popcount bisection steps instead of clearing n bits; the same
  arithmetic here against the plain definition."""
  rng = np.random.RandomState(21)
  for _ in range(2000):
    cols = int(rng.randint(0, 2 ** 32)) | (int(rng.randint(0, 2 ** 32)) << 32)
    cols &= (1 << int(rng.randint(1, 65))) - 1
    if not cols:
      continue
    n = bin(cols).count('1')
    pick = int(rng.randint(n))
    want = [c for c in range(64) if (cols >> c) & 1][pick]
    col, v, k = 0, cols, pick
    for w in (32, 16, 8, 4, 2, 1):
      below = bin(v & ((1 << w) - 1)).count('1')
      if k >= below:
        k -= below
        col += w
        v >>= w
    assert col == want, (hex(cols), pick, col, want)
