#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Harvests the reference's own known-answer tests.

Runs google-deepmind/pycolab's unchanged test modules
    pycolab/tests/maze_walker_test.py   (:33-569, three tests)
    pycolab/tests/scrolling_test.py     (:112-504, Scrolly x MazeWalker, both margin modes)
    pycolab/tests/cropping_test.py      (:75-654, seven cropper tests)
from /root/reference with `PycolabTestCase.assertMachinima` wrapped, and turns
every machinima into a fixture under tests/golden/reftests/:

  * the game: the arguments of the test's own `ascii_art.ascii_art_to_game`
    call, rebuilt with pycolab_amd's ascii_art and its tabled prefabs
    (`tt.TestMazeWalker` -> `tabled.TabledWalker`, `tt.TestScrolly` ->
    `tabled.TabledScrolly`, do-nothing entities -> static programs) and
    compiled to a `GameTemplate`; start-of-game teleports that the tests inject
    with `tt.pre_update` are applied to the twin's entities before compiling;
  * the actions: the test's strings / dicts of compass directions, packed into
    one integer per frame (four bits per entity, 8 = stay);
  * the EXPECTED observations: the ASCII-art diagrams written in the reference
    test source, parsed with the reference's own `ascii_art_to_uint8_nparray`
    (for cropper tests: one diagram per cropper, plus the cropper
    constructors' arguments);
  * for testNotConfinedToBoard the expected (position, virtual_position) pairs.

The wrapped assertMachinima then runs the original, so the reference checks
itself while being harvested.  tests/test_reference_known_answers.py replays
the fixtures on the CPU oracle and, on a GPU, through the HIP engine.

Run here (CPU container):  python oracle/harvest_reference_tests.py
"""
import json
import os
import sys

if os.environ.get('PYTHONHASHSEED') != '0':
  # games built from sets of characters iterate them in string-hash order: pin it so
  # that the fixtures come out byte-identical on every run (tests/test_fixture_reproducibility.py)
  os.environ['PYTHONHASHSEED'] = '0'
  os.execv(sys.executable, [sys.executable] + sys.argv)
import unittest
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('PCX_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)
sys.dont_write_bytecode = True
warnings.filterwarnings('ignore', category=DeprecationWarning)

from pycolab import ascii_art as ref_art  # noqa: E402
from pycolab import cropping as ref_cropping  # noqa: E402
from pycolab.tests import test_things as tt  # noqa: E402

from pycolab_amd import ascii_art as our_art  # noqa: E402
from pycolab_amd.compiler import GameTemplate  # noqa: E402
from pycolab_amd.prefab_parts import tabled  # noqa: E402

MOTIONS = ['n', 'ne', 'e', 'se', 's', 'sw', 'w', 'nw']
OUT = os.path.join(os.environ.get('PCX_GOLDEN_OUT') or os.path.join(ROOT, 'tests', 'golden'), 'reftests')
FIXTURES = []


def twin_of(args, kwargs, pre_updates):
  """The same game built with pycolab_amd's tabled prefabs; returns (template,
  {character: action field index})."""
  kw = dict(kwargs)
  names = ['art', 'what_lies_beneath', 'sprites', 'drapes', 'backdrop', 'update_schedule', 'z_order']
  for name, value in zip(names, args):
    kw[name] = value
  fields = {}

  def convert(partial, is_sprite, ch):
    cls, a, k = partial, (), {}
    if isinstance(partial, ref_art.Partial):
      cls, a, k = partial.pycolab_thing, partial.args, dict(partial.kwargs)
    if cls is tt.TestMazeWalker or cls is tt.TestScrolly:
      fields[ch] = len(fields)
      k['action_field'] = (4 * fields[ch], 15)
      return our_art.Partial(tabled.TabledWalker if cls is tt.TestMazeWalker else tabled.TabledScrolly, *a, **k)
    if cls.__name__ == 'DoNothingDrape' or cls is tt.TestDrape:
      return our_art.Partial(tabled.StaticDrape, *a, **k)
    if cls is tt.TestSprite:
      return our_art.Partial(tabled.StaticSprite, *a, **k)
    raise NotImplementedError('no tabled twin for %r' % (cls,))

  for key, is_sprite in (('sprites', True), ('drapes', False)):
    if kw.get(key):
      kw[key] = {ch: convert(p, is_sprite, ch) for ch, p in sorted(kw[key].items())}
  game = our_art.ascii_art_to_game(**kw)
  for ch, fn in pre_updates:  # start-of-game teleports (scrolling_test.py:180-185)
    fn(None, None, None, game.backdrop, game.things, game.the_plot)
  return GameTemplate.from_engine(game), fields


def encode(action, fields):
  """A test's action (None, a direction string, or {character: direction}) as
  the packed integer the tabled prefabs decode."""
  if action is None:
    return -1
  code = 0
  for ch, idx in fields.items():
    direction = action if isinstance(action, str) else (action.get(ch) if isinstance(action, dict) else None)
    code |= (MOTIONS.index(direction) if direction in MOTIONS else 8) << (4 * idx)
  return code


def cropper_spec(c):
  if c is None or type(c) is ref_cropping.ObservationCropper:
    return None
  if isinstance(c, ref_cropping.FixedCropper):
    return dict(kind='fixed', top_left=[c._top_row, c._left_col], rows=c._rows, cols=c._cols, pad_char=c._pad_char)
  assert isinstance(c, ref_cropping.ScrollingCropper)
  return dict(kind='scrolling', rows=c._rows, cols=c._cols, to_track=list(c._to_track), pad_char=c._pad_char,
              scroll_margins=list(c._scroll_margins), initial_offset=list(c._initial_offset), saccade=bool(c._saccade))


_real_a2g = ref_art.ascii_art_to_game
_real_pre_update = tt.pre_update
_real_machinima = tt.PycolabTestCase.assertMachinima


def a2g(*args, **kwargs):
  game = _real_a2g(*args, **kwargs)
  game._pcx_build = (args, kwargs)
  game._pcx_pre = []
  return game


def pre_update(engine, character, thing_to_do):
  if hasattr(engine, '_pcx_pre') and not engine._showtime:  # injected before its_showtime(): a start-of-game setup
    engine._pcx_pre.append((character, thing_to_do))
  return _real_pre_update(engine, character, thing_to_do)


def assertMachinima(self, engine, frames, pre_updates=None, post_updates=None, result_checker=None, croppers=None):
  frames = [tuple(f[:1]) + ((tuple(f[1]) if croppers is not None else f[1]),) + tuple(f[2:]) for f in frames]
  name = '%s_%d' % (self.id().split('.')[-1], len([f for f in FIXTURES if f.rsplit('_', 1)[0] == self.id().split('.')[-1]]))
  try:
    template, fields = twin_of(*engine._pcx_build, pre_updates=engine._pcx_pre)
  except NotImplementedError as e:
    print('skipped %s: %s' % (name, e))
    return _real_machinima(self, engine, frames, pre_updates, post_updates, result_checker, croppers)
  actions = np.array([encode(f[0], fields) for f in frames], np.int32)
  arrays = {'actions': actions}
  meta = dict(test=self.id(), fields=fields, frame0_is_checked=False)
  if croppers is None:
    arrays['boards'] = np.stack([ref_art.ascii_art_to_uint8_nparray(f[1]) for f in frames])
  else:
    croppers = tuple(croppers)
    meta['croppers'] = [cropper_spec(c) for c in croppers]
    # cropping_test.py:66-71 shows the first observation to the croppers before the machinima
    meta['croppers_primed'] = all(c is None or c._engine is engine for c in croppers)
    for i in range(len(croppers)):
      arrays['crop_%d' % i] = np.stack([ref_art.ascii_art_to_uint8_nparray(f[1][i]) for f in frames])
  if self.id().endswith('testNotConfinedToBoard'):  # machinima_args = (position, virtual_position) of P
    arrays['positions'] = np.array([[f[2][0], f[2][1], f[3][0], f[3][1]] for f in frames], np.int32)
  os.makedirs(OUT, exist_ok=True)
  template.save(os.path.join(OUT, name + '.template.npz'))
  arrays['meta'] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
  np.savez_compressed(os.path.join(OUT, name + '.npz'), **arrays)
  FIXTURES.append(name)
  print('wrote %s: %d frames, %s' % (name, len(frames), 'croppers %d' % len(croppers) if croppers is not None else 'boards'))
  return _real_machinima(self, engine, frames, pre_updates, post_updates, result_checker, croppers)


_real_assert_array_equal = np.testing.assert_array_equal


def assert_array_equal(actual, desired, *args, **kwargs):
  """scrolling_test.py:143-155 compares `whole_pattern` with
  `np.array([list(row) for row in ['1010...']]).astype(bool)`, which under
  numpy 2 is all True (non-empty strings): a breakage of the reference's test
  on this stack, not of Scrolly.PatternInfo.  That one comparison is skipped so
  that the test reaches its machinimas, which hold the known answers."""
  import inspect
  caller = inspect.stack()[1]
  if (caller.filename.endswith('scrolling_test.py') and caller.function == 'testScrolly' and
      getattr(desired, 'dtype', None) == np.bool_ and desired.all() and not np.asarray(actual).all()):
    print('note: skipped the numpy-2-broken whole_pattern comparison at scrolling_test.py:%d' % caller.lineno)
    return None
  return _real_assert_array_equal(actual, desired, *args, **kwargs)


def main():
  np.testing.assert_array_equal = assert_array_equal
  ref_art.ascii_art_to_game = a2g
  tt.pre_update = pre_update
  tt.PycolabTestCase.assertMachinima = assertMachinima
  from pycolab.tests import cropping_test, maze_walker_test, scrolling_test
  suite = unittest.TestSuite()
  for mod in (maze_walker_test, scrolling_test, cropping_test):
    suite.addTests(unittest.defaultTestLoader.loadTestsFromModule(mod))
  result = unittest.TextTestRunner(verbosity=1).run(suite)
  print('reference tests: run %d, failures %d, errors %d' % (result.testsRun, len(result.failures), len(result.errors)))
  for _, tb in result.failures + result.errors:
    print(tb)
  with open(os.path.join(OUT, 'INDEX.json'), 'w') as f:
    json.dump(sorted(FIXTURES), f, indent=1)
  if not result.wasSuccessful():
    sys.exit(1)


if __name__ == '__main__':
  main()
