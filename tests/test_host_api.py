"""CPU tests of the host-side API surface (no GPU, no compute calls)."""
import ctypes
import os

import pytest

from pycolab_amd import _native as N
from pycolab_amd import ascii_art, compat, engine, programs, things
from pycolab_amd.compiler import GameTemplate
from pycolab_amd.prefab_parts import drapes as prefab_drapes
from pycolab_amd.prefab_parts import sprites as prefab_sprites
from tests import helpers

REF_EXAMPLES = '/root/reference/pycolab/examples'
needs_reference = pytest.mark.skipif(not os.path.isdir(REF_EXAMPLES),
                                     reason='reference checkout not present')


def test_library_exports_every_declared_symbol():
  """libpcx.so loads and exports every symbol include/pcx.h declares."""
  lib = ctypes.CDLL(N.LIB_PATH)
  header = open(os.path.join(helpers.ROOT, 'include', 'pcx.h')).read()
  import re
  declared = set(re.findall(r'\b(pcx_[a-z0-9_]+)\s*\(', header))
  declared -= {'pcx_cropper_'}
  bound = {name for name, _, _ in N.SYMBOLS}
  assert declared == bound, (declared ^ bound)
  for name in declared:
    assert hasattr(lib, name), name
  assert N.lib().pcx_abi_version() == N.ABI_VERSION
  assert N.lib().pcx_action_hash(1, 2, 3) == __import__('oracle.binding', fromlist=['x']).action_hash(1, 2, 3)


def test_ctypes_struct_sizes_match_header():
  """Compile a probe against include/pcx.h and compare struct sizes."""
  import subprocess, tempfile
  src = ('#include "pcx.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n",'
         'sizeof(pcx_sprite_desc),sizeof(pcx_drape_desc),sizeof(pcx_template),sizeof(pcx_buffers),'
         'sizeof(pcx_sprite_state),sizeof(pcx_cropper_desc));return 0;}')
  with tempfile.TemporaryDirectory() as d:
    open(os.path.join(d, 'p.c'), 'w').write(src)
    subprocess.check_call(['gcc', '-I', os.path.join(helpers.ROOT, 'include'), '-o',
                           os.path.join(d, 'p'), os.path.join(d, 'p.c')])
    got = [int(x) for x in subprocess.check_output([os.path.join(d, 'p')]).split()]
  want = [ctypes.sizeof(c) for c in (N.SpriteDesc, N.DrapeDesc, N.Template, N.Buffers,
                                     N.SpriteState, N.CropperDesc)]
  assert got == want


def test_ascii_art_errors():
  """Restates tests/ascii_art_test.py:33-58 plus ascii_art.py:163-212 guards."""
  ascii_art.ascii_art_to_uint8_nparray(['ab', 'ba'])
  with pytest.raises(ValueError, match='except for the concatenation axis must match exactly'):
    ascii_art.ascii_art_to_uint8_nparray(['ab', 'bab'])
  with pytest.raises(TypeError, match='the argument to ascii_art_to_uint8_nparray must be a list'):
    ascii_art.ascii_art_to_uint8_nparray(['a', 2])
  with pytest.raises(TypeError, match='Did you pass a list of list of single characters?'):
    ascii_art.ascii_art_to_uint8_nparray([['a', 'b'], ['b', 'a']])
  art = ['ab', 'ba']
  two = {'a': things.Sprite, 'b': things.Sprite}
  with pytest.raises(TypeError):   # mixed nesting in update_schedule
    ascii_art.ascii_art_to_game(art, ' ', sprites=two, update_schedule=[['a'], 7])
  with pytest.raises(ValueError):  # schedule must list everything once
    ascii_art.ascii_art_to_game(art, ' ', sprites=two, update_schedule='a')
  with pytest.raises(ValueError):  # z_order must list everything once
    ascii_art.ascii_art_to_game(art, ' ', sprites=two, z_order='a')
  with pytest.raises(ValueError):  # what_lies_beneath: one char or art
    ascii_art.ascii_art_to_game(art, 'xy', sprites=two)
  with pytest.raises(ValueError):  # beneath must not be a thing's char
    ascii_art.ascii_art_to_game(art, 'a', sprites=two)
  with pytest.raises(ValueError):  # a sprite may appear once
    ascii_art.ascii_art_to_game(art, ' ', sprites=two)
  with pytest.raises(TypeError):
    ascii_art.Partial(int)


def test_engine_builder_errors():
  """engine.py:851-874 guards."""
  e = engine.Engine(3, 4)
  e.set_backdrop('. ', things.Backdrop)
  with pytest.raises(RuntimeError):
    e.set_backdrop('x', things.Backdrop)          # second backdrop
  with pytest.raises(RuntimeError):
    e.add_sprite('.', (0, 0), things.Sprite)      # claimed by backdrop
  with pytest.raises(TypeError):
    e.add_sprite('s', (0, 0), things.Drape)
  with pytest.raises(ValueError):
    e.add_sprite('s', (3, 0), things.Sprite)      # off-board
  with pytest.raises(ValueError):
    e.add_sprite('ss', (0, 0), things.Sprite)
  e.add_sprite('s', (1, 1), things.Sprite)
  with pytest.raises(ValueError):
    e.set_z_order('sx')
  with pytest.raises(RuntimeError):
    e.play(0)                                     # before its_showtime


def test_unknown_entity_class_fails_loudly():
  class Mystery(things.Sprite):
    def update(self, actions, board, layers, backdrop, things, the_plot):
      self._position = self.Position(0, 0)
  game = ascii_art.ascii_art_to_game(['m.'], '.', sprites={'m': Mystery})
  with pytest.raises(programs.UnsupportedEntityError):
    game.its_showtime()


def test_maze_walker_constructor_semantics():
  corner = things.Sprite.Position(5, 5)
  with pytest.raises(ValueError):
    prefab_sprites.MazeWalker(corner, things.Sprite.Position(1, 1), 'x', 'x#')
  w = prefab_sprites.MazeWalker(corner, things.Sprite.Position(1, 1), 'x', '#')
  w._teleport((-3, 2))
  assert w.position == (0, 0) and w.virtual_position == (-3, 2) and not w.visible
  w._teleport((2, 2))
  assert w.position == (2, 2) and w.visible and w.on_the_board
  with pytest.raises(NotImplementedError):
    w._north(None, None)


def test_palette():
  p = engine.Palette('#. _')
  assert p.hash == ord('#') and p['.'] == ord('.') and p._ == ord('_') and '#' in p
  with pytest.raises(AttributeError):
    p.at  # pylint: disable=pointless-statement
  with pytest.raises(IndexError):
    p['@']  # pylint: disable=pointless-statement


def test_template_roundtrip(tmp_path):
  t = helpers.load_template('scrolly_maze_L0')
  path = str(tmp_path / 't.npz')
  t.save(path)
  assert GameTemplate.load(path) == t
  ct, keep = t.to_ctypes()
  assert ct.rows == 10 and ct.cols == 30 and ct.n_sprites == 4 and ct.n_drapes == 2
  assert bytes(ct.z_order[:6]) == b'abc@#P' and bytes(ct.schedule[:6]) == b'#abcP@'
  assert list(ct.group_of[:6]) == [0, 1, 1, 1, 1, 2] and ct.n_groups == 3
  assert bytes(ct.chars[:ct.n_chars]) == bytes(sorted(b' #.@Pabc'))


@needs_reference
@pytest.mark.parametrize('level', [0, 1, 2])
def test_unchanged_reference_example_compiles_to_fixture(level):
  """The shipped example file, loaded unchanged with `pycolab` aliased to this
  package, builds the same template as the committed fixture."""
  mod = compat.load_game_module(os.path.join(REF_EXAMPLES, 'scrolly_maze.py'))
  game = mod.make_game(level)
  assert isinstance(game, engine.Engine)
  assert isinstance(game.things['P'], prefab_sprites.MazeWalker)
  assert isinstance(game.things['#'], prefab_drapes.Scrolly)
  t = GameTemplate.from_engine(game)
  assert t == helpers.load_template('scrolly_maze_L%d' % level)
  assert t.game == N.GAME_SCROLLY_MAZE and t.n_actions == 5


_ALL_EXAMPLES = [('scrolly_maze.py', (0,), 'scrolly_maze_L0'), ('warehouse_manager.py', (0,), 'warehouse_L0'),
                 ('warehouse_manager.py', (1,), 'warehouse_L1'), ('warehouse_manager.py', (2,), 'warehouse_L2'),
                 ('extraterrestrial_marauders.py', (), 'marauders'), ('hello_world.py', (), 'hello_world'),
                 ('better_scrolly_maze.py', (0,), 'better_scrolly_maze_L0'),
                 ('better_scrolly_maze.py', (1,), 'better_scrolly_maze_L1'),
                 ('better_scrolly_maze.py', (2,), 'better_scrolly_maze_L2')]


@needs_reference
@pytest.mark.parametrize('hashseed', ['1', '4242'])
def test_every_shipped_example_resolves_in_a_fresh_process(hashseed):
  """Class -> device program matching must not depend on the process (string
  hash seed, object addresses): each example file of the config games loads
  unchanged and compiles to its committed fixture in a new interpreter."""
  import subprocess
  import sys
  script = r'''
import os, sys
sys.path.insert(0, %r)
from pycolab_amd import compat
from pycolab_amd.compiler import GameTemplate
from tests import helpers
for fname, args, fixture in %r:
  mod = compat.load_game_module(os.path.join(%r, fname))
  t = GameTemplate.from_engine(mod.make_game(*args))
  want = helpers.load_template(fixture)
  if fixture == 'hello_world':  # its update schedule iterates a set of characters (hello_world.py:90)
    assert sorted(t.schedule) == sorted(want.schedule) and t.game == want.game, fixture
  else:
    assert t == want, fixture
print('ok')
''' % (helpers.ROOT, _ALL_EXAMPLES, REF_EXAMPLES)
  env = dict(os.environ, PYTHONHASHSEED=hashseed, PCX_NO_TORCH='1')
  out = subprocess.run([sys.executable, '-c', script], env=env, capture_output=True, text=True, timeout=300)
  assert out.returncode == 0 and out.stdout.strip().endswith('ok'), out.stderr[-2000:]


@needs_reference
def test_reference_ascii_art_test_file_passes_unchanged():
  """The reference's own tests/ascii_art_test.py, loaded unchanged with
  `pycolab` aliased to this package, passes (it is the one reference test file
  that needs no Python-side entity stepping)."""
  import unittest
  mod = compat.load_game_module('/root/reference/pycolab/tests/ascii_art_test.py')
  suite = unittest.defaultTestLoader.loadTestsFromModule(mod)
  assert suite.countTestCases() >= 1
  result = unittest.TextTestRunner(verbosity=0).run(suite)
  assert result.wasSuccessful(), result.failures + result.errors


_WALKER_SRC = '''
from pycolab_amd.prefab_parts import sprites as prefab_sprites


class PlayerSprite(prefab_sprites.MazeWalker):
  """%(doc)s"""

  def __init__(self, corner, position, character):
    super(PlayerSprite, self).__init__(corner, position, character, impassable='#')

  def update(self, actions, board, layers, backdrop, things, the_plot):%(comment)s
    if actions == 0:
      self._north(board, the_plot)
    elif actions == %(south)s:
      self._south(   board,
                     the_plot)
'''


def _load_source(tmp_path, name, **subst):
  import importlib.util
  path = tmp_path / (name + '.py')
  path.write_text(_WALKER_SRC % subst)
  spec = importlib.util.spec_from_file_location(name, str(path))
  mod = importlib.util.module_from_spec(spec)  # deliberately NOT entered into sys.modules
  spec.loader.exec_module(mod)
  return mod.PlayerSprite


def test_class_fingerprint_is_of_the_normalised_source(tmp_path):
  """programs.fingerprint hashes the class's normalised AST: docstrings,
  comments, formatting and the interpreter do not matter; the code does.  It
  is not a bytecode digest, so it does not tie the package to one CPython."""
  from pycolab_amd import programs
  a = _load_source(tmp_path, 'game_a', doc='A player.', comment='', south='1')
  b = _load_source(tmp_path, 'game_b', doc='Something else entirely.\n\n  Longer.', comment='  # a remark', south='1')
  c = _load_source(tmp_path, 'game_c', doc='A player.', comment='', south='2')
  fa, fb, fc = programs.fingerprint(a), programs.fingerprint(b), programs.fingerprint(c)
  assert fa is not None and fa == fb and fa != fc
  # an edited copy of a shipped class name must not be matched to a device program
  with pytest.raises(programs.UnsupportedEntityError) as err:
    programs.resolve(c(things.Sprite.Position(5, 5), things.Sprite.Position(1, 1), 'P'))
  assert 'edited' in str(err.value)
  # every digest in the shipped table has the AST form (16 hex digits) and no bytecode dependence
  from pycolab_amd._shipped_fingerprints import SHIPPED_FINGERPRINTS
  assert all(isinstance(fp, str) and len(fp) == 16 for _, fp in SHIPPED_FINGERPRINTS)


import numpy as np  # noqa: E402


def test_batched_story_last_assignment_of_next_chapter_wins():
  """plot.py:310-311: "the last call before termination determines what happens".  Batch > 1: the host's
  set_next_chapter() and the entities' the_plot.next_chapter assignments (device words) -- whichever spoke last; the
  same rule the batch-1 plot applies.  Host logic only: no device."""
  from pycolab_amd import _native as N
  from pycolab_amd import storytelling
  s = object.__new__(storytelling.Story)
  s._keys, s._chapters, s._auto_advance, s._batch = [0, 1, 2, 3], {0: None, 1: None, 2: None, 3: None}, True, 4
  s._next_override, s._entity_next_at_override = None, {}
  U, NONE = N.CHAPTER_UNSET, N.CHAPTER_NONE
  assigned = np.array([U, 2, 3, NONE], np.int32)
  # nobody on the host spoke: the entities' words, else the list order
  assert [s._next_of(e, 0, assigned) for e in range(4)] == [1, 2, 3, None]
  assert s._next_of(0, 3, assigned) is None  # (the last chapter of the list)
  # the host speaks when the words held [U, 2, 2, U]: environment 1's word has not changed since (the host is later),
  # environments 2 and 3 were assigned after it (the entities are later), environment 0's entities never spoke
  s._next_override = [0, 0, 0, 0]
  s._entity_next_at_override = {0: np.array([U, 2, 2, U], np.int32)}
  assert [s._next_of(e, 0, assigned) for e in range(4)] == [0, 0, 3, None]
  # a chapter whose engine had no words yet when the host spoke: any word is a later assignment
  assert [s._next_of(e, 1, assigned) for e in range(4)] == [0, 2, 3, None]
  # engines whose entities never assign (assigned is None): the host's value
  assert [s._next_of(e, 0, None) for e in range(4)] == [0, 0, 0, 0]


def test_baked_level_constants_header_is_what_the_library_plans():
  """csrc/pcx_sm_shipped.h (the shipped scrolly_maze levels' kernel constants, compiled into instances of
  pcx_scrolly_maze_step) is generated from pcx_debug_scrolly_consts and committed: it must be what the library plans for
  the golden templates today, or the fast instances silently stop being used.  Every level answers its own words."""
  import importlib.util
  spec = importlib.util.spec_from_file_location('gen_sm_shipped', os.path.join(helpers.ROOT, 'tools', 'gen_sm_shipped.py'))
  gen = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(gen)
  arrs = gen.arrays()
  assert gen.header_arrays() == arrs, 'run tools/gen_sm_shipped.py and rebuild libpcx.so'
  assert open(gen.HEADER).read() == gen.render(arrs)
  assert len(set(tuple(w) for w in arrs.values())) == len(arrs) == 6  # three levels x {persistent, plain}: all different
  words = arrs['PCX_SM_SHIPPED_L0_WORDS']
  t1 = helpers.load_template('scrolly_maze_L1')
  ct, _keep = t1.to_ctypes()
  buf = (ctypes.c_uint32 * len(words))()
  n = N.lib().pcx_debug_scrolly_consts(ctypes.byref(ct), 64, buf, len(words))
  assert n == len(words) and list(buf) != words
  wh = helpers.load_template('warehouse_L0')
  ct, _keep = wh.to_ctypes()
  assert N.lib().pcx_debug_scrolly_consts(ctypes.byref(ct), 64, None, 0) < 0  # (not a scrolly_maze template)


@pytest.mark.skipif(not os.path.isdir(REF_EXAMPLES), reason='needs the reference sources')
def test_every_reference_example_builds_on_the_host_and_the_unported_ones_fail_loudly():
  """All 18 example files of the reference load UNCHANGED against this package (pycolab_amd/compat.py) and build their
  game through the mirrored construction API; the five games of the hot path and the Story of examples/ordeal.py compile to templates, every other one
  stops at the template compiler with UnsupportedEntityError naming the entity class that has no device program --
  never a silent fallback, never a different error (out of scope, and loudly so)."""
  import glob
  import inspect
  from pycolab_amd.programs import UnsupportedEntityError
  hot_path = {'scrolly_maze.py', 'better_scrolly_maze.py', 'warehouse_manager.py', 'extraterrestrial_marauders.py', 'hello_world.py'}
  compiled, refused = set(), set()
  for path in sorted(glob.glob(os.path.join(REF_EXAMPLES, '**', '*.py'), recursive=True)):
    rel = os.path.relpath(path, REF_EXAMPLES)
    if os.path.basename(path) == '__init__.py':
      continue
    module = compat.load_game_module(path)
    make = module.make_game
    params = list(inspect.signature(make).parameters)
    args = [np.random.RandomState(0) if p in ('rng', 'random_state') else 0 for p in params]
    if rel == 'ordeal.py':
      # a Story of three games (its constructor starts every chapter once on the device, storytelling.py:556-624): the
      # chapter builders are caught on their way in and compiled one by one -- to the very templates the GPU tests step
      import types
      caught, real = {}, module.storytelling
      module.storytelling = types.SimpleNamespace(Story=lambda chapters, **kw: caught.update(chapters))
      try:
        make()
      finally:
        module.storytelling = real
      assert sorted(caught) == ['castle', 'cavern', 'kansas']
      for key, build in caught.items():
        game = build()
        game.the_plot._this_chapter = key
        t = GameTemplate.from_engine(game)
        assert t == helpers.load_template('ordeal_' + key) and t.reward_is_float and t.n_plot_words == 3
      compiled.add(rel)
      continue
    try:
      game = make(*args)
      GameTemplate.from_engine(game)
      compiled.add(rel)
    except UnsupportedEntityError as e:
      assert 'No device program' in str(e)
      refused.add(rel)
    except (TypeError, AttributeError, ValueError):
      # make_game() of a research example wants arguments this loop does not know how to make up
      assert rel.startswith('research'), rel
  assert compiled == hot_path | {'ordeal.py'}, compiled  # (round 6: SURVEY section 8 f-4's cited game has its device programs)
  assert len(refused) >= 10 and 'shockwave.py' in refused, refused


def test_engine_defaults_reach_engines_built_out_of_the_callers_sight():
  """`engine.defaults(batch=...)`: game files that build their engines inside `make_game()` -- the chapters of
  examples/ordeal.py:82-110 are built inside its Story -- cannot be `configure()`d by the caller; the block's batch size
  and device are what `Engine.__init__` starts from, nested blocks restore what was there, and `configure()` still wins."""
  from pycolab_amd import ascii_art, engine
  art = ['#####', '#P  #', '#####']
  build = lambda: ascii_art.ascii_art_to_game(art, what_lies_beneath=' ', sprites={'P': ascii_art.Partial(tabled_walker(), impassable='#')})
  assert build().batch == 1
  with engine.defaults(batch=4096, device=0):
    assert engine.current_defaults() == dict(batch=4096, device=0)
    assert build().batch == 4096
    with engine.defaults(batch=8):
      assert build().batch == 8
    assert build().batch == 4096
    assert build().configure(batch=2).batch == 2
  assert build().batch == 1 and engine.current_defaults() == dict(batch=1, device=0)
  with pytest.raises(ValueError):
    with engine.defaults(batch=0):
      pass
  assert engine.current_defaults() == dict(batch=1, device=0)


def tabled_walker():
  from pycolab_amd.prefab_parts import tabled
  return tabled.TabledWalker
