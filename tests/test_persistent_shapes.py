"""GPU parity of the persistent launch shapes of pcx_scrolly_maze_step (round 4): workgroups that stay on their
CU and draw work units, the next unit's state words prefetched into LDS by LDS-DMA (shape 1), logic / render wave
pairs handing owner-code buffers over a ring (shape 2).  Every shape must produce exactly what the oracle does
(reference: engine.py:583-639 via oracle/pcx_oracle.c), with units of 64 / 32 / 16 environments, dynamic tickets
or static round-robin, few workgroups walking many units, ragged batches, environments left alone (no auto-reset)
and externally supplied action tapes."""
import os

import numpy as np
import pytest

from oracle import binding
from tests import helpers
from tests.hip_adapter import HipAdapter

pytestmark = pytest.mark.gpu

KNOBS = ('PCX_SM_SHAPE', 'PCX_SM_UNIT', 'PCX_SM_DYNAMIC', 'PCX_SM_GRID', 'PCX_SM_PER_CU', 'PCX_COOP_BELOW', 'PCX_SM_WAVES',
         'PCX_SM_LOCK', 'PCX_SM_CODES', 'PCX_SM_TAIL', 'PCX_SM_TAIL_UNIT')


class Knobs(object):
  """Environment knobs of ScrollyMazeBackend::launch, restored on exit (they are read at every launch)."""

  def __init__(self, **kw):
    self.kw = kw

  def __enter__(self):
    self.saved = {k: os.environ.get(k) for k in set(KNOBS) | set(self.kw)}  # (every knob this block sets is restored, listed above or not)
    for k, v in self.kw.items():
      os.environ[k] = str(v)
    return self

  def __exit__(self, *exc):
    for k, v in self.saved.items():
      if v is None:
        os.environ.pop(k, None)
      else:
        os.environ[k] = v


class OracleAdapter(binding.OracleEngine):

  def read(self, name):
    return np.array(getattr(self, name))


def assert_same(hip, orc, where):
  for name in ('planes', 'reward', 'reward_set', 'discount', 'done', 'frame', 'error'):
    np.testing.assert_array_equal(hip.read(name), orc.read(name), err_msg='%s: %s' % (where, name))
  np.testing.assert_array_equal(hip.sprites(), orc.sprites(), err_msg=where + ': sprites')
  np.testing.assert_array_equal(hip.curtains(), orc.curtains(), err_msg=where + ': curtains')


def raw_shape_of(hip):
  from pycolab_amd import _native as N
  return int(N.lib().pcx_engine_launch_shape(hip.eng._native))


def shape_of(hip):
  """The launch shape; 5 is shape 3 run by the instance with the shipped level's constants compiled in."""
  s = raw_shape_of(hip)
  return 3 if s in (5, 13) else s  # (13: the same workers walking several steps of one launch)


@pytest.mark.parametrize('shape,codes,waves,lock,grid,dynamic', [(3, 1, 4, 1, 2, 1), (3, 1, 3, 2, 3, 0), (3, 1, 10, 3, 1, 1), (3, 1, 1, 0, 5, 1),
                                                          (3, 1, 6, 2, 1, 1), (3, 1, 2, 0, 3, 0)])
def test_persistent_workers_with_streaming_semaphore_match_oracle(shape, codes, waves, lock, grid, dynamic):
  """`waves` workers per workgroup of which at most `lock` stream at a time (0: no limit)."""
  t = helpers.load_template('scrolly_maze_L0')
  B, T = 2999, 120
  with Knobs(PCX_COOP_BELOW=0, PCX_SM_SHAPE=shape, PCX_SM_CODES=codes, PCX_SM_WAVES=waves, PCX_SM_LOCK=lock, PCX_SM_GRID=grid,
             PCX_SM_DYNAMIC=dynamic):
    hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
    hip.reset(); orc.reset()
    t0 = 0
    while t0 < T:
      n = 1 if t0 < 16 else 8
      hip.step_hashed(0x5EED, t0, n); orc.step_hashed(0x5EED, t0, n)
      assert shape_of(hip) == shape
      t0 += n
      assert_same(hip, orc, 'shape %d codes %d waves %d lock %d after step %d' % (shape, codes, waves, lock, t0))


@pytest.mark.parametrize('shape,unit,dynamic,grid,waves', [
    (3, 64, 1, 5, 1), (3, 64, 0, 5, 1), (3, 32, 1, 7, 1), (3, 64, 1, 4096, 1),
    (3, 64, 0, 5, 2), (3, 32, 1, 3, 4), (3, 16, 1, 2, 6), (3, 64, 1, 1, 2),
])  # (round 6: eight of round 5's twelve -- units of 64 / 32 / 16, tickets and round-robin, one to six workers, a grid larger than the batch)
def test_persistent_shapes_match_oracle(shape, unit, dynamic, grid, waves):
  """A ragged batch (not a multiple of 64, 32 or 16) walked by `grid` workgroups; hashed actions, resets included."""
  t = helpers.load_template('scrolly_maze_L0')
  B, T = 2999, 160
  with Knobs(PCX_COOP_BELOW=0, PCX_SM_SHAPE=shape, PCX_SM_UNIT=unit, PCX_SM_DYNAMIC=dynamic, PCX_SM_GRID=grid, PCX_SM_WAVES=waves):
    hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
    hip.reset(); orc.reset()
    assert_same(hip, orc, 'frame 0')
    t0 = 0
    while t0 < T:
      n = 1 if t0 < 24 else 8
      hip.step_hashed(0x5EED, t0, n); orc.step_hashed(0x5EED, t0, n)
      assert shape_of(hip) == shape, 'the launch took shape %d, not %d' % (shape_of(hip), shape)
      t0 += n
      assert_same(hip, orc, 'shape %d unit %d after step %d' % (shape, unit, t0))
    assert int(orc.read('frame').min()) < T  # episodes ended and restarted inside the run


@pytest.mark.parametrize('shape,tail,small', [(3, 2, 16)])  # (round 6: one setting -- the knob is off by default, every setting measured slower)
def test_small_units_at_the_end_of_the_batch(shape, tail, small):
  """With tickets the batch's last environments go in small units (so that what the workers hold when the tickets run out
  is short): 20,000 environments on four workgroups -- a few dozen 64-environment units, then units of `small`."""
  t = helpers.load_template('scrolly_maze_L0')
  B, T = 20000 + 13, 48
  with Knobs(PCX_COOP_BELOW=0, PCX_SM_SHAPE=shape, PCX_SM_GRID=4, PCX_SM_DYNAMIC=1, PCX_SM_TAIL=tail, PCX_SM_TAIL_UNIT=small):
    hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
    hip.reset(); orc.reset()
    for t0 in range(0, T, 8):
      hip.step_hashed(0x5EED, t0, 8); orc.step_hashed(0x5EED, t0, 8)
      assert shape_of(hip) == shape
      assert_same(hip, orc, 'shape %d tail %d x %d after step %d' % (shape, tail, small, t0 + 8))


@pytest.mark.parametrize('shape', [3])
def test_persistent_shapes_tape_actions_and_environments_left_alone(shape):
  """Action tapes from the host (illegal and quit actions among them), and steps without auto-reset: finished
  environments are skipped (their units stream fewer planes: the wait-for-the-prefetch path)."""
  t = helpers.load_template('scrolly_maze_L1')
  B, T = 1500, 120
  rng = np.random.RandomState(7)
  with Knobs(PCX_COOP_BELOW=0, PCX_SM_SHAPE=shape, PCX_SM_GRID=4):
    hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
    hip.reset(); orc.reset()
    for step in range(T):
      a = rng.randint(0, 5, size=B).astype(np.int32)
      r = rng.rand(B)
      a[r < 0.04] = -1
      a[(r >= 0.04) & (r < 0.06)] = 5
      a[(r >= 0.06) & (r < 0.08)] = rng.randint(6, 40)
      auto = step % 3 != 0
      hip.step(a, auto_reset=auto); orc.step(a, auto_reset=auto)
      assert shape_of(hip) == shape
      assert_same(hip, orc, 'shape %d step %d' % (shape, step))


@pytest.mark.parametrize('shape,unit', [(3, 64), (3, 32)])
def test_persistent_shapes_at_config_5_shard_size(shape, unit):
  """131,072 environments (BASELINE config 5's per-GPU shard) at the default residency: the first and the last
  2,048 environments against the oracle, layer == (board == c) over the whole batch, and everything equal to what
  the one-workgroup-per-group shape writes for the same tape."""
  import torch
  t = helpers.load_template('scrolly_maze_L0')
  B, T, K = 131072, 40, 2048
  with Knobs(PCX_SM_SHAPE=shape, PCX_SM_UNIT=unit):
    hip = HipAdapter(t, B)
    hip.reset()
    hip.step_hashed(0xC0FFEE, 0, T)
    assert shape_of(hip) == shape
    planes = hip.eng.planes_view()
  with Knobs(PCX_SM_SHAPE=0):
    ref = HipAdapter(t, B)
    ref.reset()
    ref.step_hashed(0xC0FFEE, 0, T)
    assert shape_of(ref) == 0
  assert torch.equal(planes, ref.eng.planes_view())
  for name in ('reward', 'reward_set', 'discount', 'done', 'frame', 'error'):
    assert torch.equal(hip.eng.buffers[name].tensor, ref.eng.buffers[name].tensor), name
  head, tail = OracleAdapter(t, K), OracleAdapter(t, K)
  head.reset(); tail.reset()
  head.step_hashed(0xC0FFEE, 0, T); tail.step_hashed(0xC0FFEE, 0, T, env_offset=B - K)
  np.testing.assert_array_equal(planes[:K].cpu().numpy(), head.read('planes'))
  np.testing.assert_array_equal(planes[B - K:].cpu().numpy(), tail.read('planes'))
  for name in ('reward', 'reward_set', 'discount', 'done', 'frame'):
    got = hip.eng.buffers[name].tensor
    np.testing.assert_array_equal(got[:K].cpu().numpy(), head.read(name))
    np.testing.assert_array_equal(got[B - K:].cpu().numpy(), tail.read(name))
  chars = torch.tensor(list(t.chars), dtype=torch.uint8, device=planes.device)
  want = (planes[:, :1] == chars.view(1, -1, 1, 1)).to(torch.uint8)
  assert torch.equal(planes[:, 1:], want)
  assert not hip.eng.buffers['error'].tensor.any()


@pytest.mark.parametrize('baked,waves,lock,grid,dynamic', [(1, 2, 1, 5, 1), (0, 2, 1, 5, 1), (1, 3, 2, 3, 0), (1, 6, 2, 1, 1), (1, 2, 0, 4096, 1)])
def test_instance_with_the_shipped_levels_constants_compiled_in(baked, waves, lock, grid, dynamic):
  """Round 5: shape 3 exists twice -- Consts from the kernel arguments, and the shipped level 0's Consts as compile-time
  constants (csrc/pcx_sm_shipped.h); an engine whose own constants equal the header's runs the second (launch shape 5),
  PCX_SM_BAKED=0 keeps the first.  Both against the oracle on a ragged batch, resets included."""
  t = helpers.load_template('scrolly_maze_L0')
  B, T = 2999, 160
  with Knobs(PCX_COOP_BELOW=0, PCX_SM_SHAPE=3, PCX_SM_BAKED=baked, PCX_SM_WAVES=waves, PCX_SM_LOCK=lock, PCX_SM_GRID=grid, PCX_SM_DYNAMIC=dynamic):
    hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
    hip.reset(); orc.reset()
    assert_same(hip, orc, 'frame 0')
    t0 = 0
    while t0 < T:
      n = 1 if t0 < 24 else 8
      hip.step_hashed(0x5EED, t0, n); orc.step_hashed(0x5EED, t0, n)
      assert raw_shape_of(hip) == (13 if n > 1 else 5 if baked else 3)
      t0 += n
      assert_same(hip, orc, 'baked %d after step %d' % (baked, t0))
    assert int(orc.read('frame').min()) < T


@pytest.mark.parametrize('baked', [1, 0])
def test_shipped_level_1_has_its_own_compiled_in_instance(baked):
  """Level 1 has the shipped shape and its own constants: its own compiled-in instances (persistent: launch shape 5 / 13),
  PCX_SM_BAKED=0 keeps the run-time instance; both against the oracle."""
  t = helpers.load_template('scrolly_maze_L1')
  B = 1500
  with Knobs(PCX_COOP_BELOW=0, PCX_SM_SHAPE=3, PCX_SM_GRID=4, PCX_SM_BAKED=baked):
    hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
    hip.reset(); orc.reset()
    for t0, n in ((0, 1), (1, 1), (2, 6), (8, 8), (16, 48)):
      hip.step_hashed(0x5EED, t0, n); orc.step_hashed(0x5EED, t0, n)
      assert raw_shape_of(hip) == (13 if n > 1 else 5 if baked else 3)
      assert_same(hip, orc, 'level 1 after step %d' % (t0 + n))


def test_a_level_of_ones_own_keeps_the_run_time_constants():
  """The shipped level 0 with one patroller moved: the same shape, other constants -- no compiled-in instance may take it."""
  t = helpers.load_template('scrolly_maze_L0')
  t.sprites[0]['vcol'] += 2
  B = 1500
  with Knobs(PCX_COOP_BELOW=0, PCX_SM_SHAPE=3, PCX_SM_GRID=4):
    hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
    hip.reset(); orc.reset()
    for t0 in range(0, 48, 1):
      hip.step_hashed(0x5EED, t0, 1); orc.step_hashed(0x5EED, t0, 1)
      assert raw_shape_of(hip) == 3
    assert_same(hip, orc, 'after 48 steps')
  with Knobs(PCX_SM_BAKED=1):  # ... nor the cooperative one (small batch, default knobs): results are the oracle's either way
    hip, orc = HipAdapter(t, 700), OracleAdapter(t, 700)
    hip.reset(); orc.reset()
    hip.step_hashed(0x5EED, 0, 40); orc.step_hashed(0x5EED, 0, 40)
    assert_same(hip, orc, 'cooperative shape')


def test_headline_batch_default_shape_equals_one_workgroup_per_group_shape_everywhere():
  """1,048,576 environments (the headline): what the default launch shape writes -- persistent workers, LDS-DMA state
  prefetch that relies on vmcnt retiring in order, the baked-constants instance -- equals, over the WHOLE batch, what the
  one-workgroup-per-group shape writes for the same tape; the first, the last and 2,048 environments from the middle
  against the oracle."""
  import torch
  t = helpers.load_template('scrolly_maze_L0')
  B, T, K = 1048576, 24, 2048
  hip = HipAdapter(t, B)
  hip.reset()
  for t0 in range(T // 2):  # single-step launches: the shape Engine.play() takes ...
    hip.step_hashed(0xC0FFEE, t0, 1)
    assert raw_shape_of(hip) in (3, 5)  # (5: the constants-compiled-in instance; as tests/test_gate_digests.py)
  # ... then one launch of twelve steps, every worker on its own units (PCX_FUSE_STEPS=2: above ~650,000 environments step_n
  # issues single-step launches by itself -- tickets and stealing win there, profiles/r06_stepn_crossover.txt)
  with Knobs(PCX_FUSE_STEPS=2):
    hip.step_hashed(0xC0FFEE, T // 2, T - T // 2)
  assert raw_shape_of(hip) == 13
  planes = hip.eng.planes_view()
  with Knobs(PCX_SM_SHAPE=0):
    ref = HipAdapter(t, B)
    ref.reset()
    ref.step_hashed(0xC0FFEE, 0, T)
    assert raw_shape_of(ref) == 0
  step = 65536
  for lo in range(0, B, step):  # (piecewise: no 2.8 GB temporary)
    assert torch.equal(planes[lo:lo + step], ref.eng.planes_view()[lo:lo + step]), 'planes differ in environments %d..' % lo
  for name in ('reward', 'reward_set', 'discount', 'done', 'frame', 'error'):
    assert torch.equal(hip.eng.buffers[name].tensor, ref.eng.buffers[name].tensor), name
  del ref
  for off in (0, B // 2 - 1000, B - K):
    orc = OracleAdapter(t, K)
    orc.reset()
    orc.step_hashed(0xC0FFEE, 0, T, env_offset=off)
    np.testing.assert_array_equal(planes[off:off + K].cpu().numpy(), orc.read('planes'), err_msg='environments from %d' % off)
    for name in ('reward', 'reward_set', 'discount', 'done', 'frame'):
      np.testing.assert_array_equal(hip.eng.buffers[name].tensor[off:off + K].cpu().numpy(), orc.read(name), err_msg=name)
  assert not hip.eng.buffers['error'].tensor.any()


@pytest.mark.parametrize('baked', [1, 0])
@pytest.mark.parametrize('B,level', [(250, 0), (4096, 0), (16391, 0), (3000, 1), (2000, 2)])
def test_cooperative_shape_with_the_levels_constants_compiled_in(B, level, baked):
  """Small batches (BASELINE config 2): the cooperative instance also exists with the shipped level's constants compiled in
  (PCX_SM_BAKED=0: the instance that reads them from the kernel arguments); single launches and launches of several steps,
  16 / 32 / 64 environments per workgroup, against the oracle."""
  t = helpers.load_template('scrolly_maze_L%d' % level)
  with Knobs(PCX_SM_BAKED=baked):
    hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
    hip.reset(); orc.reset()
    t0 = 0
    for n in ((1, 1, 1, 7, 64, 3, 1, 120) if B < 10000 else (1, 1, 7, 33, 1)):  # (the largest batch: fewer steps, the oracle is what it waits for)
      hip.step_hashed(0x5EED, t0, n); orc.step_hashed(0x5EED, t0, n)
      if 'PCX_COOP_BELOW' not in os.environ or os.environ['PCX_COOP_BELOW'] not in ('0',):  # (a suite run with the cooperative shape forced off compares results only)
        assert raw_shape_of(hip) == (12 if n > 1 else 10)
      t0 += n
      assert_same(hip, orc, 'B %d level %d baked %d after step %d' % (B, level, baked, t0))
    assert level == 2 or B > 10000 or int(orc.read('frame').min()) < t0  # (level 2's patrollers are boxed in: nobody is caught within 198 steps)


# ---- the kernels built on pcx_stream.h: persistent workers (round 5) ------------------------------------------------------

STREAM_KERNELS = [('PCX_WM_', 'warehouse_L0', 5), ('PCX_WM_', 'warehouse_L1', 5), ('PCX_WM_', 'warehouse_L2', 5), ('PCX_WM_', 'warehouse_custom_A', 5),
                  ('PCX_WM_', 'warehouse_custom_B', 5), ('PCX_HW_', 'hello_world', 4), ('PCX_HW_', 'hello_custom_A', 4)]


# every worker shape on the two shipped games, two of them on the other levels (the shapes differ in scheduling only)
STREAM_CASES = [(p, n, o, w) for p, n, o in STREAM_KERNELS for i, w in enumerate([(8, 4, 0, 1), (3, 1, 1, 2), (1, 0, 0, 4), (4, 2, 1, 3)])
                if n in ('warehouse_L0', 'hello_world') or i in (1, 3)]


@pytest.mark.parametrize('prefix,name,n_ordinary,shape', STREAM_CASES)
def test_stream_kernels_persistent_workers_match_oracle(prefix, name, n_ordinary, shape):
  """pcx_warehouse_step / pcx_hello_world_step with persistent workers (launch shape 3) against the oracle: a ragged batch
  that every worker walks several units of (a handful of workgroups: PCX_xx_GRID), hashed actions and host tapes with
  quirky actions, steps without auto-reset."""
  from pycolab_amd import _native as N
  workers, lock, dynamic, grid = shape
  t = helpers.load_template(name)
  B, T = 64 * 37 + 13, 64
  rng = np.random.RandomState(5)
  with Knobs(**{'PCX_COOP_BELOW': 0, prefix + 'WORKERS': workers, prefix + 'LOCK': lock, prefix + 'DYNAMIC': dynamic, prefix + 'GRID': grid}):
    hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
    hip.reset(); orc.reset()
    assert_same(hip, orc, 'frame 0')
    t0 = 0
    while t0 < T:
      n = 1 if t0 < 8 else 8
      hip.step_hashed(0x5EED, t0, n); orc.step_hashed(0x5EED, t0, n)
      assert int(N.lib().pcx_engine_launch_shape(hip.eng._native)) == 3
      t0 += n
      assert_same(hip, orc, '%s workers %d lock %d after step %d' % (name, workers, lock, t0))
    for step in range(12):
      a = rng.randint(0, n_ordinary, size=B).astype(np.int32)
      r = rng.rand(B)
      a[r < 0.04] = -1
      a[(r >= 0.04) & (r < 0.06)] = n_ordinary
      a[(r >= 0.06) & (r < 0.08)] = rng.randint(n_ordinary + 1, 40)
      auto = step % 3 != 0
      hip.step(a, auto_reset=auto); orc.step(a, auto_reset=auto)
      assert int(N.lib().pcx_engine_launch_shape(hip.eng._native)) == 3
      assert_same(hip, orc, '%s tape step %d' % (name, step))


@pytest.mark.parametrize('prefix,name,B', [('PCX_HW_', 'hello_world', 262144)])
def test_stream_kernels_persistent_workers_equal_the_round_2_shape(prefix, name, B):
  import torch
  from pycolab_amd import _native as N
  t = helpers.load_template(name)
  T = 32
  hip = HipAdapter(t, B)
  hip.reset(); hip.step_hashed(0xC0FFEE, 0, T)
  assert int(N.lib().pcx_engine_launch_shape(hip.eng._native)) == 3
  with Knobs(**{prefix + 'PW': 0}):
    ref = HipAdapter(t, B)
    ref.reset(); ref.step_hashed(0xC0FFEE, 0, T)
    assert int(N.lib().pcx_engine_launch_shape(ref.eng._native)) == 0
  assert torch.equal(hip.eng.planes_view(), ref.eng.planes_view())
  for key in ('reward', 'reward_set', 'discount', 'done', 'frame', 'error'):
    assert torch.equal(hip.eng.buffers[key].tensor, ref.eng.buffers[key].tensor), key


def test_warehouse_persistent_workers_equal_the_round_2_shape_at_config_4():
  """262,144 environments (BASELINE config 4): persistent workers == one workgroup per group over the whole batch, and
  layer == (board == c) everywhere."""
  import torch
  from pycolab_amd import _native as N
  t = helpers.load_template('warehouse_L0')
  B, T = 262144, 48
  hip = HipAdapter(t, B)
  hip.reset(); hip.step_hashed(0xC0FFEE, 0, T)
  assert int(N.lib().pcx_engine_launch_shape(hip.eng._native)) == 3
  with Knobs(PCX_WM_PW=0):
    ref = HipAdapter(t, B)
    ref.reset(); ref.step_hashed(0xC0FFEE, 0, T)
    assert int(N.lib().pcx_engine_launch_shape(ref.eng._native)) == 0
  planes = hip.eng.planes_view()
  assert torch.equal(planes, ref.eng.planes_view())
  for name in ('reward', 'reward_set', 'discount', 'done', 'frame', 'error'):
    assert torch.equal(hip.eng.buffers[name].tensor, ref.eng.buffers[name].tensor), name
  chars = torch.tensor(list(t.chars), dtype=torch.uint8, device=planes.device)
  assert torch.equal(planes[:, 1:], (planes[:, :1] == chars.view(1, -1, 1, 1)).to(torch.uint8))
  K = 2048
  for off in (0, B - K):
    orc = OracleAdapter(t, K)
    orc.reset(); orc.step_hashed(0xC0FFEE, 0, T, env_offset=off)
    np.testing.assert_array_equal(planes[off:off + K].cpu().numpy(), orc.read('planes'))


@pytest.mark.parametrize('baked,waves,grid', [(1, 4, 2), (0, 2, 3), (1, 1, 5), (1, 3, 4096)])
def test_persistent_workers_walk_several_steps_per_launch(baked, waves, grid):
  """Round 5: `step_n` / `step_hashed` at large batches are launches of up to 64 steps in which every persistent worker keeps
  its own units from step to step (launch shape 13: no ramp and no tail between the steps; the state words round-trip
  through HBM and come back by LDS-DMA, waited for where a worker wraps around to a unit it has just written).  Against the
  oracle: hashed actions and host tapes with quirky actions, chunks that leave finished environments frozen, a grid of one
  unit per worker (every step wraps onto the unit just written) and of many."""
  t = helpers.load_template('scrolly_maze_L0')
  B = 64 * 23 + 13
  rng = np.random.RandomState(3)
  with Knobs(PCX_COOP_BELOW=0, PCX_SM_SHAPE=3, PCX_SM_BAKED=baked, PCX_SM_WAVES=waves, PCX_SM_GRID=grid):
    hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
    hip.reset(); orc.reset()
    t0 = 0
    for i, n in enumerate((1, 2, 7, 64, 100, 3, 1, 31)):
      auto = i % 3 != 2
      hip.step_hashed(0xABCD, t0, n, auto_reset=auto); orc.step_hashed(0xABCD, t0, n, auto_reset=auto)
      assert raw_shape_of(hip) == (13 if n > 1 else (5 if baked else 3)), (n, raw_shape_of(hip))
      t0 += n
      assert_same(hip, orc, 'baked %d waves %d grid %d after %d steps (chunk %d)' % (baked, waves, grid, t0, n))
    tape = rng.randint(0, 5, size=(70, B)).astype(np.int32)
    r = rng.rand(70, B)
    tape[r < 0.03] = -1
    tape[(r >= 0.03) & (r < 0.04)] = 5
    tape[(r >= 0.04) & (r < 0.06)] = 17
    hip.eng._auto_reset = True
    hip.eng.step_n(tape)
    for row in tape:
      orc.step(row)
    assert raw_shape_of(hip) == 13
    assert_same(hip, orc, 'after the tape')
