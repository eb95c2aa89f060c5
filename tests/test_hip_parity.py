"""GPU parity tests: the HIP step path (through the C ABI) against
(1) golden traces recorded from the imported reference, and
(2) the CPU oracle on seeded inputs, and
(3) size-independent properties at the BASELINE batch sizes."""
import os

import numpy as np
import pytest

from oracle import binding
from tests import helpers
from tests.hip_adapter import HipAdapter

pytestmark = pytest.mark.gpu

LEVELS = ['scrolly_maze_L0', 'scrolly_maze_L1', 'scrolly_maze_L2']
ALL_GAMES = LEVELS + ['warehouse_L0', 'warehouse_L1', 'warehouse_L2', 'hello_world', 'marauders',
                      'scrolly_maze_L1_unoccluded', 'warehouse_L0_unoccluded', 'marauders_unoccluded',
                      'walkers_room', 'walkers_hidden', 'walkers_scroll_margins', 'walkers_scroll_always', 'walkers_scroll_groups',
                      'better_scrolly_maze_L0', 'better_scrolly_maze_L1', 'better_scrolly_maze_L2',
                      # the step kernel's shape-generic instances (oracle/custom_levels.py)
                      'scrolly_custom_A', 'scrolly_custom_B', 'scrolly_custom_C', 'scrolly_custom_D', 'scrolly_custom_E',
                      'scrolly_custom_A_unoccluded', 'scrolly_custom_C_unoccluded', 'scrolly_custom_E_unoccluded',
                      'scrolly_custom_F', 'scrolly_custom_G', 'scrolly_custom_H', 'warehouse_custom_A', 'warehouse_custom_B', 'marauders_custom_A', 'hello_custom_A',
                      # run-time-shape instances of pcx_warehouse_step / pcx_better_scrolly_step
                      'warehouse_custom_C', 'warehouse_custom_D', 'better_scrolly_custom_A', 'better_scrolly_custom_B', 'better_scrolly_custom_C', 'better_scrolly_custom_D', 'better_scrolly_custom_E',
                      # Plot directives incl. change_z_order on the device (engine.py:796-835)
                      'directives_z_order', 'directives_reward_discount', 'directives_two_discounts', 'directives_float_rewards']


class OracleAdapter(binding.OracleEngine):

  def read(self, name):
    return np.array(getattr(self, name))


def assert_same(hip, orc, where):
  for name in ('planes', 'reward', 'reward_set', 'discount', 'done', 'frame', 'error'):
    np.testing.assert_array_equal(hip.read(name), orc.read(name), err_msg='%s: %s' % (where, name))
  np.testing.assert_array_equal(hip.sprites(), orc.sprites(), err_msg=where + ': sprites')
  np.testing.assert_array_equal(hip.curtains(), orc.curtains(), err_msg=where + ': curtains')


@pytest.mark.parametrize('name', ALL_GAMES)
def test_hip_matches_reference_trace(name):
  helpers.replay_trace(HipAdapter, helpers.load_trace(name))


@pytest.mark.parametrize('name', ALL_GAMES)
def test_hip_matches_oracle_hashed_actions(name):
  """2048 envs x 192 steps (the shipped scrolly_maze levels; fewer for the others), uniform actions from the shared counter hash,
  resets included; every output compared every 8 steps and at the end."""
  t = helpers.load_template(name)
  t.param[0] = 0xBEEF  # RNG seed (marauders)
  # (round 6: sizes halved -- the oracle is what the test waits for; every launch shape of every kernel has tests of its own)
  B, T = (2048, 192) if name in LEVELS else (512, 96) if name.startswith('better') else (768, 128) if name.startswith('marauders') else (1024, 144)
  hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
  hip.reset(); orc.reset()
  assert_same(hip, orc, 'frame 0')
  resets = 0
  # the first 32 steps (and 16 more in the middle) are compared after every
  # single step -- a transient wrong observation cannot hide between checks --
  # the rest every 8 steps
  t0 = 0
  while t0 < T:
    n = 1 if (t0 < 32 or T // 2 <= t0 < T // 2 + 16) else 8
    hip.step_hashed(0x5EED, t0, n); orc.step_hashed(0x5EED, t0, n)
    t0 += n
    assert_same(hip, orc, 'after step %d' % t0)
    resets += int(orc.read('done').sum())
  assert resets > 0 or not name.startswith(('scrolly_maze_L0', 'scrolly_maze_L1', 'marauders'))  # reset path exercised (L2 patrollers are boxed in)


@pytest.mark.parametrize('name,T', [('scrolly_maze_L0', 4000), ('scrolly_maze_L2', 2000), ('warehouse_L0', 3000),
                                    ('marauders', 2500), ('hello_world', 1500), ('better_scrolly_maze_L1', 1200)])
def test_long_run_matches_oracle(name, T):
  """Thousands of consecutive steps per environment (many episodes end and
  restart, frame counters and RNG draw counters run far beyond what the
  traces reach): everything compared at checkpoints, state included."""
  t = helpers.load_template(name)
  t.param[0] = 0xFACE
  B = 192
  hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
  hip.reset(); orc.reset()
  episodes = 0
  for t0 in range(0, T, 250):
    n = min(250, T - t0)
    hip.step_hashed(0xABCDE, t0, n); orc.step_hashed(0xABCDE, t0, n)
    assert_same(hip, orc, '%s after step %d' % (name, t0 + n))
    episodes += int(orc.read('done').sum())
  assert int(orc.read('frame').max()) > 0


@pytest.mark.parametrize('name', LEVELS)
def test_hip_matches_oracle_single_wave_launch_shape(name, monkeypatch):
  """Small batches take the cooperative launch shape (four waves share a
  group's render loop); BASELINE-size batches take single-wave workgroups.
  Force the latter at test size and compare with the oracle as above."""
  monkeypatch.setenv('PCX_COOP_BELOW', '0')
  t = helpers.load_template(name)
  B, T = 4096, 128
  hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
  hip.reset(); orc.reset()
  assert_same(hip, orc, 'frame 0')
  for t0 in range(0, T, 16):
    hip.step_hashed(0xC0FFEE, t0, 16); orc.step_hashed(0xC0FFEE, t0, 16)
    assert_same(hip, orc, 'after step %d' % (t0 + 16))


@pytest.mark.parametrize('build', helpers.BUILDS)
@pytest.mark.parametrize('waves', ['1', '2', '4'])
@pytest.mark.parametrize('name', ['marauders', 'warehouse_L2', 'walkers_scroll_groups', 'directives_z_order', 'hello_world',
                                  'warehouse_L0_unoccluded'])
def test_hip_table_driven_kernel_waves_per_workgroup(name, waves, build, monkeypatch):
  """The table-driven kernel picks 2, 4 or 8 waves per workgroup from the
  batch and the LDS footprint; test-size batches always get 8.  Force the
  others (what BASELINE-size batches run) and compare with the oracle."""
  from pycolab_amd import _native as N
  helpers.force_generic(monkeypatch, build)
  monkeypatch.setenv('PCX_GENERIC_WAVES', waves)
  t = helpers.load_template(name)
  t.param[0] = 0xD1CE
  B = 64 * 10 + 37
  hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
  hip.reset(); orc.reset()
  assert N.lib().pcx_engine_kernel_name(hip.eng._native).decode() == 'pcx_generic_step'
  for t0 in range(0, 96, 16):
    hip.step_hashed(0x5EED, t0, 16); orc.step_hashed(0x5EED, t0, 16)
    assert_same(hip, orc, 'after step %d' % (t0 + 16))


@pytest.mark.parametrize('name,knob,value', [('marauders', 'PCX_EM_WAVES', '1'), ('marauders', 'PCX_EM_WAVES', '8'),
                                             ('warehouse_L0', 'PCX_COOP_BELOW', '0'), ('warehouse_L2', 'PCX_COOP_BELOW', '0'),
                                             ('warehouse_custom_B', 'PCX_COOP_BELOW', '0')])
def test_hip_hand_written_kernels_other_launch_shapes(name, knob, value, monkeypatch):
  """pcx_marauders_step / pcx_warehouse_step pick their launch shape from the
  batch (test-size batches get four waves per group); force the single-wave
  shape that BASELINE-size batches run, and the eight-wave one, and compare
  with the oracle after every step."""
  monkeypatch.setenv(knob, value)
  t = helpers.load_template(name)
  t.param[0] = 0xA11CE
  B = 200
  hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
  hip.reset(); orc.reset()
  assert_same(hip, orc, 'frame 0')
  for t0 in range(80):
    hip.step_hashed(0x5EED, t0, 1); orc.step_hashed(0x5EED, t0, 1)
    assert_same(hip, orc, 'after step %d' % (t0 + 1))
  from pycolab_amd import _native as N
  assert N.lib().pcx_engine_kernel_name(hip.eng._native).decode() in ('pcx_marauders_step', 'pcx_warehouse_step')


@pytest.mark.parametrize('name,kernel', [('better_scrolly_maze_L0', 'pcx_better_scrolly_step'), ('better_scrolly_maze_L1', 'pcx_better_scrolly_step'),
                                         ('better_scrolly_maze_L2', 'pcx_better_scrolly_step'), ('marauders', 'pcx_marauders_step'),
                                         ('warehouse_L1', 'pcx_warehouse_step'), ('scrolly_maze_L2', 'pcx_scrolly_maze_step'),
                                         ('marauders_custom_A', 'pcx_generic_step'), ('hello_world', 'pcx_hello_world_step'),
                                         ('hello_custom_A', 'pcx_hello_world_step'),
                                         ('warehouse_custom_C', 'pcx_warehouse_step'), ('warehouse_custom_D', 'pcx_warehouse_step'),
                                         ('better_scrolly_custom_A', 'pcx_better_scrolly_step'), ('better_scrolly_custom_B', 'pcx_better_scrolly_step'),
                                         # unwalled boards: things off the board at (0, 0), patrollers that look around it from outside
                                         ('better_scrolly_custom_D', 'pcx_better_scrolly_step'), ('better_scrolly_custom_E', 'pcx_better_scrolly_step'),
                                         # occlusion_in_layers=False: the warehouse kernel's own UNOCC instances (round 3);
                                         # marauders' rules read layers, so its unoccluded variant stays table-driven
                                         ('warehouse_L0_unoccluded', 'pcx_warehouse_step'), ('marauders_unoccluded', 'pcx_generic_step')])
@pytest.mark.parametrize('shape', ['coop', 'single'])
def test_which_kernel_steps_which_game(name, kernel, shape, monkeypatch):
  """The shipped shapes run the hand-written kernels, everything else the table-driven
  one; both launch shapes of each (several waves per group at small batches, single-wave
  workgroups at BASELINE sizes) match the oracle step by step, quirky actions included."""
  if shape == 'single':
    monkeypatch.setenv('PCX_COOP_BELOW', '0')
    monkeypatch.setenv('PCX_EM_WAVES', '1')
  from pycolab_amd import _native as N
  t = helpers.load_template(name)
  t.param[0] = 0xD00D
  B = 150
  hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
  hip.reset(); orc.reset()
  assert N.lib().pcx_engine_kernel_name(hip.eng._native).decode() == kernel
  rng = np.random.RandomState(8)
  for step in range(48):
    a = rng.randint(0, t.n_actions, size=B).astype(np.int32)
    r = rng.rand(B)
    a[r < 0.03] = -1
    a[(r >= 0.03) & (r < 0.04)] = t.n_actions   # the quit action of every shipped game
    hip.step(a, auto_reset=step % 5 != 4); orc.step(a, auto_reset=step % 5 != 4)
    assert_same(hip, orc, '%s step %d' % (name, step))


@pytest.mark.parametrize('build', helpers.BUILDS)
@pytest.mark.parametrize('name', ['better_scrolly_custom_D', 'better_scrolly_custom_E'])
def test_unwalled_better_scrolly_traces_through_the_table_driven_kernel(name, build, monkeypatch):
  """prog_bs_patroller (pcx_generic_kernel.h) on the reference's own recording of patrollers that leave an unwalled board:
  out there they look around position (0, 0) -- in custom_E with a wall on both sides of it, where the SECOND test of
  better_scrolly_maze.py:291-294 wins and they walk west for good (test_which_kernel_steps_which_game and
  test_hip_matches_reference_trace hold pcx_better_scrolly_step to the same traces)."""
  from pycolab_amd import _native as N
  helpers.force_generic(monkeypatch, build)
  eng = helpers.replay_trace(HipAdapter, helpers.load_trace(name))
  assert N.lib().pcx_engine_kernel_name(eng.eng._native).decode() == 'pcx_generic_step'


@pytest.mark.parametrize('shape', ['coop', 'single'])
@pytest.mark.parametrize('name,kernel', [('hello_world', 'pcx_hello_world_step'), ('hello_custom_A', 'pcx_hello_world_step'),
                                         ('warehouse_L2', 'pcx_warehouse_step'), ('warehouse_custom_C', 'pcx_warehouse_step')])
def test_unoccluded_layers_from_the_hand_written_kernels(name, kernel, shape, monkeypatch):
  """Engine(..., occlusion_in_layers=False) (rendering.py:187-301) on games whose rules read no layer an
  unoccluded renderer changes: the hand-written kernels' UNOCC instances, against the oracle's unoccluded
  renderer -- raw drape curtains, visible sprites' own cells, the backdrop's characters where the backdrop has them."""
  if shape == 'single':
    monkeypatch.setenv('PCX_COOP_BELOW', '0')
  from pycolab_amd import _native as N
  t = helpers.load_template(name)
  t.occlusion_in_layers = False
  B = 200
  hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
  hip.reset(); orc.reset()
  assert N.lib().pcx_engine_kernel_name(hip.eng._native).decode() == kernel
  rng = np.random.RandomState(18)
  differs = False
  for step in range(40):
    a = rng.randint(0, t.n_actions, size=B).astype(np.int32)
    hip.step(a, auto_reset=step % 5 != 4); orc.step(a, auto_reset=step % 5 != 4)
    assert_same(hip, orc, '%s step %d' % (name, step))
    planes = hip.read('planes')
    differs |= bool((planes[:, 1:] != (planes[:, :1] == np.array(list(t.chars), np.uint8)[None, :, None, None])).any())
  assert differs, 'no layer ever differed from board == char: the unoccluded renderer was not exercised'


@pytest.mark.parametrize('fuse', ['0', '1'])
def test_hip_step_n_fused_and_unfused_match_oracle(fuse, monkeypatch):
  """pcx_engine_step_n / _step_hashed take several steps per launch at small
  batches (logic wave one step ahead of the render waves); PCX_FUSE_STEPS=0 is
  the one-launch-per-step path.  Both against the oracle, odd chunk lengths."""
  monkeypatch.setenv('PCX_FUSE_STEPS', fuse)
  t = helpers.load_template('scrolly_maze_L0')
  B = 1000
  hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
  hip.reset(); orc.reset()
  t0 = 0
  for i, n in enumerate((1, 2, 3, 7, 64, 5, 300, 40)):
    auto = i % 3 != 2  # every third chunk leaves finished environments frozen
    hip.step_hashed(0xABCD, t0, n, auto_reset=auto); orc.step_hashed(0xABCD, t0, n, auto_reset=auto)
    t0 += n
    assert_same(hip, orc, 'after %d steps (chunk %d)' % (t0, n))
  rng = np.random.RandomState(11)
  tape = rng.randint(0, 5, size=(37, B)).astype(np.int32)
  hip.eng.step_n(tape)
  for row in tape:
    orc.step(row, auto_reset=True)
  assert_same(hip, orc, 'after the tape')


@pytest.mark.parametrize('name,B', [('scrolly_maze_L0', 250), ('scrolly_maze_L0', 4096), ('scrolly_maze_L0', 16384 + 7),
                                    ('scrolly_maze_L1', 3000), ('scrolly_maze_L2', 2000)])
def test_hip_several_steps_per_launch_in_the_cooperative_shape(name, B):
  """Round 4: a launch of several steps (pcx_engine_step_n / _step_hashed at small batches) is the cooperative
  instance walking the steps with the state words in registers -- 16, 32 or 64 environments per workgroup, four lanes
  per environment at the small end, six coin words (level 2: the words past the fourth stay in LDS) -- against the
  oracle: odd chunk lengths, chunks that leave finished environments frozen, explicit tapes with quirky actions."""
  from pycolab_amd import _native as N
  t = helpers.load_template(name)
  hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
  hip.reset(); orc.reset()
  t0 = 0
  for i, n in enumerate((1, 2, 3, 9, 64, 5, 200, 31) if B < 10000 else (1, 2, 9, 40, 5)):  # (the largest batch: fewer steps, the oracle is what it waits for)
    auto = i % 3 != 2
    hip.step_hashed(0xABCD, t0, n, auto_reset=auto); orc.step_hashed(0xABCD, t0, n, auto_reset=auto)
    if n > 1 and 'PCX_COOP_BELOW' not in os.environ:  # (a suite run with the cooperative shape forced off / on compares results only)
      assert int(N.lib().pcx_engine_launch_shape(hip.eng._native)) == 12
    t0 += n
    assert_same(hip, orc, '%s x %d after %d steps (chunk %d)' % (name, B, t0, n))
  rng = np.random.RandomState(11)
  tape = rng.randint(0, 5, size=(41, B)).astype(np.int32)
  r = rng.rand(41, B)
  tape[r < 0.03] = -1
  tape[(r >= 0.03) & (r < 0.04)] = 5
  tape[(r >= 0.04) & (r < 0.06)] = 17
  for auto in (True, False):
    hip.eng._auto_reset = auto
    hip.eng.step_n(tape)
    for row in tape:
      orc.step(row, auto_reset=auto)
    assert_same(hip, orc, '%s x %d after the tape (auto_reset %s)' % (name, B, auto))


@pytest.mark.parametrize('name', ['scrolly_maze_L1', 'scrolly_custom_B', 'warehouse_L1', 'marauders', 'hello_world',
                                  'better_scrolly_maze_L1', 'walkers_scroll_margins', 'marauders_unoccluded'])
def test_hip_matches_oracle_quirky_actions(name):
  """Explicit action tapes with None / quit / out-of-range actions and no
  auto-reset (finished environments stay frozen)."""
  t = helpers.load_template(name)
  t.param[0] = 0xFACE  # RNG seed (marauders)
  B, T = (512, 160) if name == 'scrolly_maze_L1' else (192, 96)
  rng = np.random.RandomState(3)
  hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
  hip.reset(); orc.reset()
  for step in range(T):
    a = rng.randint(0, 5, size=B).astype(np.int32)
    r = rng.rand(B)
    a[r < 0.04] = -1
    a[(r >= 0.04) & (r < 0.05)] = 5
    a[(r >= 0.05) & (r < 0.08)] = rng.randint(6, 40)
    auto = step % 3 != 0
    hip.step(a, auto_reset=auto); orc.step(a, auto_reset=auto)
    assert_same(hip, orc, 'step %d' % step)


def test_full_batch_properties():
  """BASELINE size (scrolly_maze L0, 1,048,576 envs on one GPU): shard
  invariance against a small oracle run, and layer/board consistency."""
  import torch
  t = helpers.load_template('scrolly_maze_L0')
  B, T, K = 1 << 20, 48, 2048
  hip = HipAdapter(t, B)
  orc = OracleAdapter(t, K)
  hip.reset(); orc.reset()
  hip.step_hashed(0xC0FFEE, 0, T); orc.step_hashed(0xC0FFEE, 0, T)
  assert hip.eng.planes.tensor is not None and hip.eng.planes.tensor.is_cuda
  planes = hip.eng.planes_view()
  # environments are independent: the first K of the big batch == a K batch
  np.testing.assert_array_equal(planes[:K].cpu().numpy(), orc.read('planes'))
  for name in ('reward', 'reward_set', 'discount', 'done', 'frame'):
    np.testing.assert_array_equal(hip.eng.buffers[name].tensor[:K].cpu().numpy(), orc.read(name))
  # ... and the last K of the big batch == a K batch offset by B-K
  orc2 = OracleAdapter(t, K)
  orc2.reset(); orc2.step_hashed(0xC0FFEE, 0, T, env_offset=B - K)
  np.testing.assert_array_equal(planes[B - K:].cpu().numpy(), orc2.read('planes'))
  # occluded layers partition the board: exactly one layer set per cell, and
  # layer k is set exactly where board == chars[k] (rendering.py:177-179)
  board = planes[:, 0]
  chars = torch.tensor(list(t.chars), dtype=torch.uint8, device=planes.device)
  for lo in range(0, B, 1 << 17):
    chunk = planes[lo:lo + (1 << 17)]
    want = (chunk[:, :1] == chars.view(1, -1, 1, 1)).to(torch.uint8)
    assert torch.equal(chunk[:, 1:], want)
  assert not hip.eng.buffers['error'].tensor.any()
  del board


@pytest.mark.parametrize('name,B,T', [('marauders', 32768, 160), ('warehouse_L0', 262144, 64)])
def test_full_batch_properties_configs_3_and_4(name, B, T):
  """BASELINE configs 3 and 4 at their full batch sizes (marauders 32,768;
  warehouse_manager level 0 262,144): the first and the last K environments
  against the oracle (with the matching env_offset, which also feeds the
  marauders' RNG draws), layer == (board == c) over the whole batch, and no
  error bits, checked at several points of the run (episodes end and restart
  inside it)."""
  import torch
  t = helpers.load_template(name)
  t.param[0] = 0xBEEF
  K = 1024
  hip = HipAdapter(t, B)
  head = OracleAdapter(t, K)
  tt = helpers.load_template(name)
  tt.param[0] = 0xBEEF
  tt.param[2], tt.param[3] = (B - K) & 0xFFFFFFFF, (B - K) >> 32  # global index of the tail's first environment
  tail = OracleAdapter(tt, K)
  hip.reset(); head.reset(); tail.reset()
  chars = torch.tensor(list(t.chars), dtype=torch.uint8, device='cuda')
  done_seen = 0
  for t0 in range(0, T, T // 4):
    n = T // 4
    hip.step_hashed(0xC0FFEE, t0, n); head.step_hashed(0xC0FFEE, t0, n)
    tail.step_hashed(0xC0FFEE, t0, n, env_offset=B - K)
    planes = hip.eng.planes_view()
    where = '%s after step %d' % (name, t0 + n)
    np.testing.assert_array_equal(planes[:K].cpu().numpy(), head.read('planes'), err_msg=where + ' (head)')
    np.testing.assert_array_equal(planes[B - K:].cpu().numpy(), tail.read('planes'), err_msg=where + ' (tail)')
    for key in ('reward', 'reward_set', 'discount', 'done', 'frame', 'error'):
      got = hip.eng.buffers[key].tensor
      np.testing.assert_array_equal(got[:K].cpu().numpy(), head.read(key), err_msg=where + ' head ' + key)
      np.testing.assert_array_equal(got[B - K:].cpu().numpy(), tail.read(key), err_msg=where + ' tail ' + key)
    for lo in range(0, B, 1 << 15):
      chunk = planes[lo:lo + (1 << 15)]
      want = (chunk[:, :1] == chars.view(1, -1, 1, 1)).to(torch.uint8)
      assert torch.equal(chunk[:, 1:], want), where
    assert not hip.eng.buffers['error'].tensor.any(), where
    done_seen += int(hip.eng.buffers['done'].tensor.sum().item())
  assert done_seen > 0 or name != 'marauders'  # marauders episodes end (and restart) inside the run; random play never solves a warehouse


@pytest.mark.parametrize('name,B,T,generic', [('better_scrolly_maze_L0', 139264, 24, False), ('scrolly_maze_L0', 1703936, 24, False),
                                              ('marauders', 720896, 24, False), ('hello_world', 1245184, 12, False),
                                              ('warehouse_L0', 3407872, 12, False), ('hello_world', 1245184, 6, True)])
def test_observation_planes_beyond_4_gib(name, B, T, generic, monkeypatch):
  """Maximum sizes: batches whose observation planes exceed 4 GiB (byte offsets
  no longer fit 32 bits; the GPU holds 288 GB).  The last K environments live
  beyond the 4 GiB mark: they and the first K against the oracle, layer ==
  (board == c) over a slice that straddles the mark, no error bits."""
  import torch
  if generic:  # the table-driven kernel
    monkeypatch.setenv('PCX_FORCE_GENERIC', '1')
  t = helpers.load_template(name)
  t.param[0] = 0xBEEF
  K = 512
  hip = HipAdapter(t, B)
  head = OracleAdapter(t, K)
  tt = helpers.load_template(name)
  tt.param[0] = 0xBEEF
  tt.param[2], tt.param[3] = (B - K) & 0xFFFFFFFF, (B - K) >> 32
  tail = OracleAdapter(tt, K)
  hip.reset(); head.reset(); tail.reset()
  hip.step_hashed(0xC0FFEE, 0, T); head.step_hashed(0xC0FFEE, 0, T); tail.step_hashed(0xC0FFEE, 0, T, env_offset=B - K)
  planes = hip.eng.planes_view()
  assert planes.numel() > (1 << 32), 'the case must exceed 4 GiB of planes'
  from pycolab_amd import _native as N
  assert (N.lib().pcx_engine_kernel_name(hip.eng._native).decode() == 'pcx_generic_step') == generic
  np.testing.assert_array_equal(planes[:K].cpu().numpy(), head.read('planes'), err_msg=name + ' (head)')
  np.testing.assert_array_equal(planes[B - K:].cpu().numpy(), tail.read('planes'), err_msg=name + ' (tail)')
  for key in ('reward', 'reward_set', 'discount', 'done', 'frame', 'error'):
    got = hip.eng.buffers[key].tensor
    np.testing.assert_array_equal(got[:K].cpu().numpy(), head.read(key), err_msg=name + ' head ' + key)
    np.testing.assert_array_equal(got[B - K:].cpu().numpy(), tail.read(key), err_msg=name + ' tail ' + key)
  per_env = planes[0].numel()
  mark = (1 << 32) // per_env  # the environment the 4 GiB mark falls into
  chars = torch.tensor(list(t.chars), dtype=torch.uint8, device='cuda')
  for lo in (0, mark - 2048, B - 4096):
    chunk = planes[lo:lo + 4096]
    want = (chunk[:, :1] == chars.view(1, -1, 1, 1)).to(torch.uint8)
    assert torch.equal(chunk[:, 1:], want), '%s environments %d..' % (name, lo)
  assert not hip.eng.buffers['error'].tensor.any()
  hip.eng.close()


@pytest.mark.parametrize('name,e', [('scrolly_maze_L0', 5), ('hello_world', 3), ('hello_world', 11), ('warehouse_L0', 2), ('marauders', 7)])
def test_engine_facade_batch1_matches_trace(name, e):
  """`Engine.play()` with batch 1 -- BASELINE config 1's shape; hello_world is the game it names -- returns the reference's
  types and values: one environment of the reference's recorded trace, step by step."""
  tr = helpers.load_trace(name)
  t = helpers.load_template(name)
  from pycolab_amd.engine import Engine
  eng = Engine.from_template(t, batch=1, seed=helpers.GOLDEN_RNG_SEED, env_offset=e)  # (marauders draws by global environment index)
  obs, reward, discount = eng.its_showtime()
  np.testing.assert_array_equal(obs.board, tr['boards'][0, e])
  want_r0 = int(tr['reward'][0, e]) if tr['reward_set'][0, e] else None
  assert reward == want_r0 and discount == 1.0 and obs.board.shape == (t.rows, t.cols)
  assert obs.layers[chr(t.chars[1])].dtype == np.bool_ and isinstance(obs.board, np.ndarray)
  for step in range(tr['actions'].shape[0]):
    if eng.game_over:
      with pytest.raises(RuntimeError):
        eng.play(0)
      break
    a = int(tr['actions'][step, e])
    obs, reward, discount = eng.play(None if a < 0 else a)
    np.testing.assert_array_equal(obs.board, tr['boards'][step + 1, e])
    want_r = int(tr['reward'][step + 1, e]) if tr['reward_set'][step + 1, e] else None
    assert reward == want_r and discount == float(tr['discount'][step + 1, e])
    for c in t.chars:
      np.testing.assert_array_equal(obs.layers[chr(c)], obs.board == c)
    assert eng.the_plot.frame == step + 1


def test_step_n_tape_equals_single_steps():
  """pcx_engine_step_n over a [T, B] tape == T calls of pcx_engine_step."""
  t = helpers.load_template('scrolly_maze_L1')
  B, T = 300, 40   # batch not a multiple of 64: the tail wave is partly idle
  rng = np.random.RandomState(11)
  tape = rng.randint(0, 5, size=(T, B)).astype(np.int32)
  a, b, orc = HipAdapter(t, B), HipAdapter(t, B), OracleAdapter(t, B)
  a.reset(); b.reset(); orc.reset()
  a.eng._auto_reset = True
  a.eng.step_n(tape)
  for step in range(T):
    b.step(tape[step]); orc.step(tape[step])
  assert_same(a, orc, 'step_n')
  assert_same(b, orc, 'single steps')


@pytest.mark.parametrize('name', ['scrolly_maze_L1', 'marauders'])
def test_partial_reset_and_things_views(name):
  """pcx_engine_reset with a mask restarts only the chosen environments; the
  facade's live `things` views agree with the oracle's entity state."""
  t = helpers.load_template(name)
  t.param[0] = 99
  B = 200
  hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
  hip.reset(); orc.reset()
  hip.step_hashed(5, 0, 30); orc.step_hashed(5, 0, 30)
  mask = (np.arange(B) % 3 == 0).astype(np.uint8)
  hip.eng.reset(mask); orc.reset(mask)
  assert_same(hip, orc, 'after masked reset')
  assert (hip.read('frame')[mask == 1] == 0).all()
  hip.step_hashed(5, 30, 20); orc.step_hashed(5, 30, 20)
  assert_same(hip, orc, 'after stepping on')
  things = hip.eng.things
  want = orc.sprites()
  for i, sp in enumerate(t.sprites):
    view = things[chr(sp['ch'])]
    pos, vis = view.position, view.visible
    for b in (0, 1, B - 1):
      assert tuple(pos[b]) == (want[b, i, 0], want[b, i, 1]) and vis[b] == bool(want[b, i, 4])
  cur = orc.curtains()
  for i, d in enumerate(t.drapes):
    np.testing.assert_array_equal(things[chr(d['ch'])].curtain.astype(np.uint8), cur[:, i])
