"""The reference's own known-answer tests, replayed.

tests/golden/reftests/ holds what oracle/harvest_reference_tests.py harvested
from the unchanged reference test modules (maze_walker_test.py:33-569,
scrolling_test.py:112-504, cropping_test.py:75-654): for every
`assertMachinima` call the game (rebuilt with pycolab_amd's tabled prefabs and
compiled to a template), the packed actions, and the EXPECTED boards -- the
ASCII-art diagrams written in the reference's test source.  The CPU oracle must
reproduce them (that pins the oracle on the reference-held vectors, not only on
differential traces), and so must the HIP engine and the device croppers.
"""
import json
import os

import numpy as np
import pytest

from oracle import binding
from pycolab_amd import cropping
from pycolab_amd.compiler import GameTemplate
from tests import helpers

DIR = os.path.join(helpers.GOLDEN, 'reftests')
NAMES = json.load(open(os.path.join(DIR, 'INDEX.json')))


def load(name):
  z = np.load(os.path.join(DIR, name + '.npz'))
  fx = {k: z[k] for k in z.files}
  fx['meta'] = json.loads(bytes(fx['meta']).decode())
  fx['template'] = GameTemplate.load(os.path.join(DIR, name + '.template.npz'))
  return fx


def show(board):
  return '\n'.join(bytes(row).decode() for row in np.asarray(board, np.uint8))


def check_board(got, want, where):
  got = np.asarray(got, np.uint8)
  assert got.shape == want.shape and np.array_equal(got, want), '%s\ngot:\n%s\nwant:\n%s' % (where, show(got), show(want))


def test_harvest_covers_the_reference_tests():
  tests = {load(n)['meta']['test'].split('.')[-1] for n in NAMES}
  assert tests == {'testBasicWalking', 'testNotConfinedToBoard', 'testConfinedToBoard', 'testScrolly',
                   'testDefaultCropper', 'testFixedCropper', 'testWeirdFixedCrops', 'testEgocentricScrolling',
                   'testScrollingSaccade', 'testScrollingMargins', 'testScrollingInitialOffset'}


@pytest.mark.parametrize('name', NAMES)
def test_oracle_reproduces_reference_known_answers(name):
  fx = load(name)
  t = fx['template']
  eng = binding.OracleEngine(t, 1)
  eng.reset()
  specs = fx['meta'].get('croppers')
  crops = None
  if specs is not None:
    crops = [None if sp is None else binding.OracleCropper(eng, cropping.cropper_from_spec(sp)) for sp in specs]
    if fx['meta']['croppers_primed']:
      for cr in crops:
        if cr is not None:
          cr.crop()
  for i, a in enumerate(fx['actions']):
    eng.step(np.array([a], np.int32), auto_reset=False)
    assert not eng.error[0], 'frame %d' % i
    if crops is None:
      check_board(eng.planes[0, 0], fx['boards'][i], '%s frame %d' % (name, i))
    else:
      for j, cr in enumerate(crops):
        if cr is None:
          got = eng.planes[0, 0]
        else:
          planes, err = cr.crop()
          assert not err.any()
          got = planes[0, 0]
        check_board(got, fx['crop_%d' % j][i], '%s frame %d cropper %d' % (name, i, j))
    if 'positions' in fx:  # maze_walker_test.py:383-391: position and virtual position of P
      sp = eng.sprites()[0, 0]
      assert tuple(sp[:4]) == tuple(fx['positions'][i]), 'frame %d: %s' % (i, sp)


@pytest.mark.gpu
@pytest.mark.parametrize('batch', [1, 3])
@pytest.mark.parametrize('name', NAMES)
def test_hip_reproduces_reference_known_answers(name, batch):
  """Through `Engine.play()` and the device croppers: batch 1 returns the
  reference's NumPy types; batch 3 (three copies of the environment) returns
  device tensors."""
  from pycolab_amd.engine import Engine
  fx = load(name)
  eng = Engine.from_template(fx['template'], batch=batch)
  obs = eng.its_showtime()[0]
  specs = fx['meta'].get('croppers')
  crops = None
  if specs is not None:
    crops = [cropping.ObservationCropper() if sp is None else cropping.cropper_from_spec(sp) for sp in specs]
    for cr in crops:
      cr.set_engine(eng)
      if fx['meta']['croppers_primed']:
        cr.crop(obs)
  pick = (lambda x: helpers.to_np(x)) if batch == 1 else (lambda x: helpers.to_np(x)[batch - 1])
  for i, a in enumerate(fx['actions']):
    obs = eng.play(None if a < 0 else (int(a) if batch == 1 else np.full((batch,), a, np.int32)))[0]
    if crops is None:
      check_board(pick(obs.board), fx['boards'][i], '%s frame %d' % (name, i))
    else:
      for j, cr in enumerate(crops):
        check_board(pick(cr.crop(obs).board), fx['crop_%d' % j][i], '%s frame %d cropper %d' % (name, i, j))
    if 'positions' in fx:
      p, v = eng.things['P'].position, eng.things['P'].virtual_position
      got = (p[0], p[1], v[0], v[1]) if batch == 1 else tuple(p[batch - 1]) + tuple(v[batch - 1])
      assert tuple(int(x) for x in got) == tuple(fx['positions'][i]), 'frame %d' % i
  eng.check_errors()
  for cr in crops or []:
    if not isinstance(cr, cropping.ObservationCropper) or type(cr) is not cropping.ObservationCropper:
      cr.check_errors()


@pytest.mark.gpu
@pytest.mark.parametrize('name', [n for n in NAMES if load(n)['meta'].get('croppers')])
def test_hip_fused_croppers_reproduce_reference_known_answers(name):
  """cropping_test.py's expected crops from croppers the table-driven step kernel
  runs itself (four at a time: the fused path's limit), drape trackers and
  priority lists included (testScrollingSaccade tracks ['P', '%'])."""
  from pycolab_amd.engine import Engine
  fx = load(name)
  specs = fx['meta']['croppers']
  assert fx['meta']['croppers_primed']
  real = [j for j, sp in enumerate(specs) if sp is not None]
  for lo in range(0, len(real), 4):
    chunk = real[lo:lo + 4]
    eng = Engine.from_template(fx['template'], batch=3)
    crops = [cropping.cropper_from_spec(specs[j]) for j in chunk]
    assert cropping.fuse_croppers(eng, crops) is None  # deferred to its_showtime(), which crops frame 0 (the test's priming crop)
    obs = eng.its_showtime()[0]
    assert all(cr._fused for cr in crops), name
    for i, a in enumerate(fx['actions']):
      obs = eng.play(None if a < 0 else np.full((3,), a, np.int32))[0]
      for j, cr in zip(chunk, crops):
        check_board(helpers.to_np(cr.crop(obs).board)[2], fx['crop_%d' % j][i], '%s frame %d cropper %d' % (name, i, j))
    eng.check_errors()
    for cr in crops:
      cr.check_errors()
    eng.close()


# ---- tests/engine_test.py:169-295, restated -----------------------------------
# The reference injects Plot calls into its test entities as Python callables
# (`tt.pre_update(engine, 'b', lambda ...: the_plot.change_z_order('b', 'c'))`);
# here an entity's update() is a device program, so the same calls are data on
# tabled entities (prefab_parts/tabled.py) selected by a field of the action.
# The games, the step sequences and the EXPECTED values below are the reference
# test's own (strings 'pyco'/'lab!'/'trousers' become the integers 5/7/11).

def _z_order_game():
  from pycolab_amd import ascii_art
  from pycolab_amd.prefab_parts import tabled
  P = ascii_art.Partial
  return ascii_art.ascii_art_to_game(  # engine_test.py:250-261
      art=['.abc.'], what_lies_beneath='.',
      sprites=dict(a=P(tabled.TabledWalker, impassable='', action_field=(0, 15)),
                   b=P(tabled.TabledWalker, impassable='', action_field=(4, 15), directive_field=(12, 3),
                       directives={1: [('change_z_order', 'b', 'c')], 2: [('change_z_order', 'b', None)]}),
                   c=P(tabled.TabledWalker, impassable='', action_field=(8, 15), directive_field=(14, 3),
                       directives={1: [('change_z_order', 'c', None)]})),
      z_order='abc')


STAY, EAST, WEST = 8, 2, 6
Z_ORDER_STEPS = [  # (action, expected board): engine_test.py:267-295
    (EAST | STAY << 4 | WEST << 8, b'..c..'),            # a and c walk onto b; z-order still 'abc'
    (STAY | STAY << 4 | STAY << 8 | 1 << 12, b'..b..'),  # b in front of c: 'acb'
    (STAY | STAY << 4 | STAY << 8 | 1 << 14, b'..b..'),  # c to the back: 'cab'
    (STAY | STAY << 4 | STAY << 8 | 2 << 12, b'..a..'),  # b to the back: 'bca'
]


def _reward_game(discount):
  from pycolab_amd import ascii_art
  from pycolab_amd.prefab_parts import tabled
  P = ascii_art.Partial
  end = ('terminate_episode',) if discount is None else ('terminate_episode', discount)
  return ascii_art.ascii_art_to_game(  # engine_test.py:196-207
      art=['.........', '...Q.R...', '.........'], what_lies_beneath='.',
      sprites=dict(Q=P(tabled.StaticSprite, directive_field=(0, 3), directives={1: [('add_reward', 5)], 2: [end]}),
                   R=P(tabled.StaticSprite, directive_field=(2, 3), directives={1: [('add_reward', 7)], 2: [('add_reward', 11)]})),
      update_schedule='QR')


def test_oracle_z_order_change_known_answer():
  """engine_test.py:244-295 testChangingZOrdering on the CPU oracle."""
  eng = binding.OracleEngine(_z_order_game().template, 1)
  eng.reset()
  for action, want in Z_ORDER_STEPS:
    eng.step(np.array([action], np.int32), auto_reset=False)
    assert bytes(eng.planes[0, 0, 0]) == want, (bytes(eng.planes[0, 0, 0]), want)


@pytest.mark.parametrize('discount', [None, 0.5])
def test_oracle_reward_and_episode_end_known_answer(discount):
  """engine_test.py:169-242 testRewardAndEpisodeEndWith{Default,Custom}Discount."""
  eng = binding.OracleEngine(_reward_game(discount).template, 1)
  eng.reset()
  assert eng.reward_set[0] == 0 and eng.discount[0] == 1.0 and not eng.done[0]   # :211-214
  eng.step(np.array([1 | 1 << 2], np.int32), auto_reset=False)
  assert eng.reward_set[0] and eng.reward[0] == 12 and eng.discount[0] == 1.0 and not eng.done[0]  # :226-229
  eng.step(np.array([2 | 2 << 2], np.int32), auto_reset=False)
  assert eng.reward[0] == 11 and eng.discount[0] == (0.0 if discount is None else 0.5) and eng.done[0]  # :239-242


@pytest.mark.gpu
def test_hip_z_order_change_known_answer():
  """engine_test.py:244-295 through Engine.play() on the device (a7)."""
  game = _z_order_game()
  game.its_showtime()
  for action, want in Z_ORDER_STEPS:
    obs, _, _ = game.play(action)
    assert bytes(obs.board[0]) == want
    for ch in 'abc.':
      np.testing.assert_array_equal(obs.layers[ch], obs.board == ord(ch))
  batched = _z_order_game().configure(batch=130)
  batched.its_showtime()
  for action, want in Z_ORDER_STEPS:
    obs, _, _ = batched.play(action)
    assert all(bytes(row[0]) == want for row in helpers.to_np(obs.board))


@pytest.mark.gpu
@pytest.mark.parametrize('discount', [None, 0.5])
def test_hip_reward_and_episode_end_known_answer(discount):
  """engine_test.py:169-242 through Engine.play(): reward None -> sum of the
  entities' rewards; terminate_episode with the default and a custom discount."""
  game = _reward_game(discount)
  _, reward, disc = game.its_showtime()
  assert reward is None and disc == 1.0 and not game.game_over
  _, reward, disc = game.play(1 | 1 << 2)
  assert reward == 12 and disc == 1.0 and not game.game_over
  _, reward, disc = game.play(2 | 2 << 2)
  assert reward == 11 and disc == (0.0 if discount is None else 0.5) and game.game_over
  with pytest.raises(RuntimeError):  # engine.py:622-624
    game.play(0)


def test_directive_validation_mirrors_the_reference():
  """plot.py:192-193 (discount range) and engine.py:804-814 (unknown characters)."""
  from pycolab_amd import ascii_art
  from pycolab_amd.prefab_parts import tabled
  P = ascii_art.Partial

  def make(calls):
    return ascii_art.ascii_art_to_game(['.ab.'], '.', sprites=dict(
        a=P(tabled.StaticSprite, directive_field=(0, 3), directives={1: calls}), b=tabled.StaticSprite)).template
  with pytest.raises(ValueError):
    make([('terminate_episode', 1.5)])
  with pytest.raises(RuntimeError):
    make([('change_z_order', 'q', 'a')])
  with pytest.raises(RuntimeError):
    make([('change_z_order', 'a', 'q')])
  assert len(make([('change_z_order', 'a', 'b'), ('add_reward', 3)]).directives) == 2
