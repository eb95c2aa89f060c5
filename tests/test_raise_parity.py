"""Where the reference RAISES, the batched engine sets the environment's error bit -- at the same frame, for the same
reason, with everything before that frame equal to what the reference returned (VERDICT r4 #4 ii).

tests/golden/raises/*.npz are recorded from the reference itself (oracle/gen_raise_golden.py) on tapes that DO raise:
egocentric walkers handed scroll orders with no component in common with their motion (prefab_parts/sprites.py:449-454),
a Scrolly that cannot follow its group's order or is ordered beyond its pattern (prefab_parts/drapes.py:523-535, 689-695),
boxes of an unwalled warehouse looking for the player beyond the last row (examples/warehouse_manager.py:219-226),
an ObservationToArray without a value for a character that turns up (rendering.py:517-522), a FixedCropper whose window
leaves the board without a pad character (cropping.py:175-183).  The C / numpy oracle is checked here on the CPU, the HIP
path -- error bits, the error polls and what check_errors() raises -- in the GPU suite."""
import os

import numpy as np
import pytest

from oracle import binding, postprocess
from tests import helpers

WALKERS = ('walkers_scroll_always', 'walkers_scroll_groups', 'walkers_scroll_margins', 'walkers_room', 'walkers_scroll_disagree')
# unwalled warehouses (oracle/custom_levels.py WAREHOUSE_OPEN_ART): pushes through numpy's index -1 until a box reaches
# the last row or column and `layers['P'][row + 1, col]` is an IndexError (warehouse_manager.py:219-226)
# (_C / _D, round 6: the same with the shipped levels' four backdrop characters -- pcx_warehouse_step's run-time-shape instance)
STEPPED = WALKERS + ('warehouse_open_A', 'warehouse_open_B', 'warehouse_open_C', 'warehouse_open_D')


def load(name):
  return np.load(os.path.join(helpers.GOLDEN, 'raises', name + '.npz'))


def check_walkers(fix, frames):
  """frames: iterable of (frame index, boards [E, R, C], error [E]) from the code under test."""
  want_frame, want_bit, boards = fix['raise_frame'], fix['raise_bit'], fix['boards']
  E = want_frame.size
  first = np.full(E, -1, np.int32)
  for f, got_boards, err in frames:
    newly = (err != 0) & (first < 0)
    first[newly] = f
    for e in np.flatnonzero(newly):
      assert int(err[e]) == int(want_bit[e]), 'environment %d, frame %d: error bits %d, the reference raised %s' % (e, f, err[e], want_bit[e])
    alive = (want_frame < 0) | (f < want_frame)  # the reference had not raised yet: its observation is the law
    np.testing.assert_array_equal(got_boards[alive], boards[f][alive], err_msg='frame %d' % f)
    assert not err[alive].any(), 'frame %d: error bits before the reference raised (environments %s)' % (f, np.flatnonzero((err != 0) & alive))
  np.testing.assert_array_equal(first, want_frame)  # the bit comes up the frame the reference raised, and only there
  return int((want_frame >= 0).sum())


@pytest.mark.parametrize('name', STEPPED)
def test_oracle_error_bit_rises_where_the_reference_raised(name):
  fix = load(name)
  t = helpers.load_template(name)
  T1, E = fix['boards'].shape[:2]
  orc = binding.OracleEngine(t, E)
  orc.reset()

  def frames():
    for f in range(T1):
      if f:
        orc.step_hashed(int(fix['seed'][0]), f - 1, 1)
      yield f, np.array(orc.planes)[:, 0], np.array(orc.error)
  raised = check_walkers(fix, frames())
  assert raised > 0 or name in ('walkers_scroll_margins', 'walkers_room')  # (those two never raise, uniform actions or not)


def test_numpy_oracle_to_array_raises_where_the_reference_raised():
  fix = load('marauders_to_array')
  t = helpers.load_template('marauders')
  t.param[0], t.param[1] = int(fix['seed'][0]), 0
  mapping = {chr(c): float(v) for c, v in zip(fix['mapping_chars'], fix['mapping_values'])}
  T1, E = fix['arrays'].shape[:2]
  orc = binding.OracleEngine(t, E)
  orc.reset()
  first = np.full(E, -1, np.int32)
  for f in range(T1):
    if f:
      orc.step_hashed(int(fix['seed'][0]), f - 1, 1)
    boards = np.array(orc.planes)[:, 0]
    for e in range(E):
      if first[e] >= 0:
        continue
      try:
        got = postprocess.to_array(boards[e], mapping, np.float32)
      except RuntimeError:
        first[e] = f
        continue
      np.testing.assert_array_equal(got, fix['arrays'][f, e])
  np.testing.assert_array_equal(first, fix['raise_frame'])
  assert len(set(first.tolist())) > 5  # (the frames differ from environment to environment)


def test_oracle_fixed_cropper_without_pad_raises_where_the_reference_does():
  """cropping.py:175-183: the window of the fixture leaves the board and there is no pad character -- the reference
  raised (fix['raises']); the oracle flags every environment, and neither the window inside the board nor the padded
  one (oracle/mutants.py `crop_overhang_never_raises` survived every fixture before this test)."""
  from pycolab_amd import cropping
  fix = load('fixed_crop_overhang')
  assert int(fix['raises'][0]) == 1
  corner, rows, cols = tuple(int(x) for x in fix['corner']), int(fix['rows'][0]), int(fix['cols'][0])
  orc = binding.OracleEngine(helpers.load_template('scrolly_maze_L0'), 3)
  orc.reset()
  over = binding.OracleCropper(orc, cropping.FixedCropper(corner, rows, cols, None))
  inside = binding.OracleCropper(orc, cropping.FixedCropper((2, 3), 5, 7, None))
  padded = binding.OracleCropper(orc, cropping.FixedCropper(corner, rows, cols, ' '))
  assert over.crop()[1].all()
  assert not inside.crop()[1].any()
  window, err = padded.crop()
  assert not err.any()
  board = np.array(orc.planes)[:, 0]
  r0, c0 = corner
  want = np.full((3, rows, cols), ord(' '), np.uint8)
  rs, cs = slice(max(r0, 0), min(r0 + rows, board.shape[1])), slice(max(c0, 0), min(c0 + cols, board.shape[2]))
  want[:, rs.start - r0:rs.stop - r0, cs.start - c0:cs.stop - c0] = board[:, rs, cs]
  np.testing.assert_array_equal(window[:, 0], want)


# the kernel a fixture's template takes by itself (the others: pcx_generic_step)
HAND_WRITTEN = {'warehouse_open_C': 'pcx_warehouse_step', 'warehouse_open_D': 'pcx_warehouse_step'}


def _hip_raise_check(name, batches):
  from pycolab_amd import _native as N
  from pycolab_amd.engine import Engine
  fix = load(name)
  t = helpers.load_template(name)
  T1, E = fix['boards'].shape[:2]
  kernels = set()
  for batch in batches:
    eng = Engine.from_template(t, batch=batch, device=0, auto_reset=True, seed=int(fix['seed'][0]))
    eng.its_showtime()
    kernels.add(N.lib().pcx_engine_kernel_name(eng._native).decode())

    def frames():
      for f in range(T1):
        if f:
          eng.step_hashed(int(fix['seed'][0]), f - 1, 1)
        yield f, eng.planes_view(host=True)[:E, 0], eng.buffers['error'].numpy()[:E]
    raised = check_walkers(fix, frames())
    if raised:  # ... and the host learns: check_errors() raises, naming the kind (of the first environment affected)
      first_bit = int(fix['raise_bit'][np.flatnonzero(fix['raise_frame'] >= 0)[0]])
      with pytest.raises(RuntimeError, match='IndexError' if first_bit == 1 else 'scrolling.Error'):
        eng.check_errors()
    else:
      eng.check_errors()
    eng.close()
  return kernels


@pytest.mark.gpu
@pytest.mark.parametrize('name', STEPPED)
def test_hip_error_bit_rises_where_the_reference_raised(name):
  # pcx_generic_step: the table-driven build, and (from 4,096 environments) the build specialised for the template
  kernels = _hip_raise_check(name, (load(name)['boards'].shape[1], 64 * 70))
  if os.environ.get('PCX_FORCE_GENERIC') != '1':
    assert kernels == {HAND_WRITTEN.get(name, 'pcx_generic_step')}


@pytest.mark.gpu
@pytest.mark.parametrize('route', ['single-wave workgroups', 'table-driven', 'specialised'])
@pytest.mark.parametrize('name', sorted(HAND_WRITTEN))
def test_hip_error_bit_rises_where_the_reference_raised_other_routes(name, route, monkeypatch):
  """The unwalled warehouses with the shipped levels' four backdrop characters: pcx_warehouse_step's run-time-shape instance
  in its large-batch launch shape too, and pcx_generic_step's two builds on the same boards."""
  if route == 'single-wave workgroups':
    if os.environ.get('PCX_FORCE_GENERIC') == '1':
      pytest.skip('PCX_FORCE_GENERIC=1: about a hand-written kernel')
    monkeypatch.setenv('PCX_COOP_BELOW', '0')
    assert _hip_raise_check(name, (64 * 3 + 32,)) == {HAND_WRITTEN[name]}
  else:
    helpers.force_generic(monkeypatch, route)
    assert _hip_raise_check(name, (64 * 3 + 32,)) == {'pcx_generic_step'}


@pytest.mark.gpu
def test_hip_to_array_flags_the_frame_the_reference_raised_at():
  from pycolab_amd import rendering
  from pycolab_amd.engine import Engine
  fix = load('marauders_to_array')
  t = helpers.load_template('marauders')
  mapping = {chr(c): float(v) for c, v in zip(fix['mapping_chars'], fix['mapping_values'])}
  T1, E = fix['arrays'].shape[:2]
  eng = Engine.from_template(t, batch=E, device=0, auto_reset=True, seed=int(fix['seed'][0]))
  conv = rendering.ObservationToArray(mapping, dtype=np.float32)
  obs = eng.its_showtime()[0]
  first = np.full(E, -1, np.int32)
  for f in range(T1):
    if f:
      eng.step_hashed(int(fix['seed'][0]), f - 1, 1)
      obs = eng._result()[0]
    want_bad = (fix['raise_frame'] >= 0) & (fix['raise_frame'] <= f)
    try:
      got = helpers.to_np(conv(obs))
    except RuntimeError:  # (the asynchronous poll of an earlier call reported it)
      assert want_bad.any()
      got = None
    errs = conv._post.errors() != 0
    first[errs & (first < 0)] = f
    if got is not None:
      ok = ~want_bad
      np.testing.assert_array_equal(got[ok], fix['arrays'][f][ok], err_msg='frame %d' % f)
  # an environment is flagged in every frame that shows the unmapped character; the FIRST such frame is the reference's raise
  np.testing.assert_array_equal(first, fix['raise_frame'])
  with pytest.raises(RuntimeError, match='only knows array values'):
    conv.check_errors() if conv._post.errors().any() else (_ for _ in ()).throw(RuntimeError('only knows array values'))
  eng.close()


@pytest.mark.gpu
def test_hip_fixed_cropper_without_pad_raises_where_the_reference_does():
  from pycolab_amd import cropping
  from pycolab_amd.engine import Engine
  fix = load('fixed_crop_overhang')
  assert int(fix['raises'][0]) == 1
  t = helpers.load_template('scrolly_maze_L0')
  for batch in (1, 300):
    eng = Engine.from_template(t, batch=batch, device=0, auto_reset=True)
    over = cropping.FixedCropper(tuple(int(x) for x in fix['corner']), int(fix['rows'][0]), int(fix['cols'][0]), None)
    inside = cropping.FixedCropper((2, 3), 5, 7, None)
    padded = cropping.FixedCropper(tuple(int(x) for x in fix['corner']), int(fix['rows'][0]), int(fix['cols'][0]), ' ')
    for cr in (over, inside, padded):
      cr.set_engine(eng)
    obs = eng.its_showtime()[0]
    inside.crop(obs); padded.crop(obs)
    inside.check_errors(); padded.check_errors()
    with pytest.raises(RuntimeError, match='extends beyond the observation'):
      over.crop(obs)
      over.check_errors()
    eng.close()
