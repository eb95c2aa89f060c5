"""The reference itself, stepped LIVE next to the code under test (VERDICT r4 #3b).

oracle/ref_live.py imports google-deepmind/pycolab -- from /root/reference in the build container, from the bytecode
that oracle/make_ref.py compiled into oracle/_ref everywhere else (build() makes it; it travels to the GPU box with the
other build products) -- and runs the unchanged example games on the bench's tape with the bench's reset policy.  The C
oracle (CPU suite) and Engine.step_hashed on the GPU (GPU suite) must return what it returns, array for array: board,
reward and is-None, discount, game_over.  Skipped, with the reason, only where neither copy of the reference exists."""
import numpy as np
import pytest

from oracle import binding, ref_live
from tests import helpers

needs_reference = pytest.mark.skipif(ref_live.reference_path() is None,
                                     reason='no reference here: neither /root/reference nor oracle/_ref (python oracle/make_ref.py)')

CASES = [('scrolly_maze_L0', 1000000, 24, 160), ('warehouse_L0', 262000, 16, 120), ('marauders', 32000, 12, 200),
         ('better_scrolly_maze_L0', 65000, 8, 120), ('hello_world', 5, 8, 60)]


@needs_reference
def test_the_built_reference_is_bytecode_of_the_reference_and_nothing_of_ours():
  import os
  path = ref_live.reference_path()
  assert os.path.isdir(os.path.join(path, 'pycolab', 'examples'))
  if path.endswith('_ref'):  # sourceless: no reference source text under the repository
    for _, _, files in os.walk(path):
      assert not [f for f in files if f.endswith('.py')]


@needs_reference
@pytest.mark.parametrize('name,off,n,steps', CASES)
def test_oracle_equals_the_live_reference(name, off, n, steps):
  ref = ref_live.run(name, off, n, steps)
  t = helpers.load_template(name)
  t.param[0], t.param[1], t.param[2], t.param[3] = 0x5EED, 0, off & 0xFFFFFFFF, off >> 32  # RNG seed, global index of environment 0
  orc = binding.OracleEngine(t, n)
  orc.reset()
  for f in range(steps + 1):
    if f:
      orc.step_hashed(0x5EED, f - 1, 1, env_offset=off)
    np.testing.assert_array_equal(np.array(orc.planes)[:, 0], ref['boards'][f], err_msg='%s board, frame %d' % (name, f))
    for key in ('reward', 'reward_set', 'discount', 'done'):
      np.testing.assert_array_equal(np.array(getattr(orc, key)), ref[key][f], err_msg='%s %s, frame %d' % (name, key, f))
  if name in ('scrolly_maze_L0', 'marauders', 'better_scrolly_maze_L0'):
    assert ref['done'].sum() > 0  # episodes ended and restarted on the way


@pytest.mark.gpu
@needs_reference
@pytest.mark.parametrize('name,off,n,steps', [(c[0], c[1], 4 * c[2], c[3]) for c in CASES])
def test_hip_equals_the_live_reference(name, off, n, steps):
  from pycolab_amd.engine import Engine
  ref = ref_live.run(name, off, n, steps)
  t = helpers.load_template(name)
  eng = Engine.from_template(t, batch=n, device=0, auto_reset=True, seed=0x5EED, env_offset=off)
  eng.its_showtime()
  chars = np.array(list(t.chars), np.uint8)
  for f in range(steps + 1):
    if f:
      eng.step_hashed(0x5EED, f - 1, 1, env_offset=off)
    planes = eng.planes_view(host=True)
    np.testing.assert_array_equal(planes[:, 0], ref['boards'][f], err_msg='%s board, frame %d' % (name, f))
    np.testing.assert_array_equal(planes[:, 1:], (planes[:, :1] == chars[None, :, None, None]).astype(np.uint8))  # rendering.py:177-179
    for key in ('reward', 'reward_set', 'discount', 'done'):
      np.testing.assert_array_equal(eng.buffers[key].numpy(), ref[key][f], err_msg='%s %s, frame %d' % (name, key, f))
  assert not eng.buffers['error'].numpy().any()
  eng.close()
