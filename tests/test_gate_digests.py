"""Gate-size parity against the reference itself (>= 4,096 sampled environments x >= 256 steps,
resets included, bit-exact against the imported reference on identical action sequences).

tests/golden/digests/<game>.npz (oracle/gen_digests.py) hold digests of what the IMPORTED reference returned for
environments [0, 4096) and [1,044,480, 1,048,576) of the headline batch over 256 steps of the bench's tape.  The C
oracle reproduces a sample of them in the CPU suite and all of them next to the GPU; the HIP path reproduces all of them
in the GPU suite -- scrolly_maze inside a real 1,048,576-environment engine in its default launch shape."""
import hashlib
import os

import numpy as np
import pytest

from oracle import binding, ref_live
from tests import helpers

GAMES = ('scrolly_maze_L0', 'warehouse_L0', 'marauders', 'better_scrolly_maze_L0')


def load(name):
  return np.load(os.path.join(helpers.ROOT, 'tests', 'golden', 'digests', name + '.npz'))


def frame_digests(boards, reward, reward_set, discount, done):
  """Full 32-byte chunk digests [chunks, 32] of ONE frame of n environments."""
  return ref_live.chunk_digests(boards[None], reward[None], reward_set[None], discount[None], done[None])[0]


def check_frame(fix, tag, t, chunk32, first_chunk=0, whole=True):
  want = fix['chunks_' + tag][t, first_chunk:first_chunk + chunk32.shape[0]]
  got = chunk32[:, :8]
  bad = np.flatnonzero((got != want).any(axis=1))
  assert bad.size == 0, 'frame %d (%s): chunks %s of 256 environments differ from the reference' % (t, tag, (bad + first_chunk).tolist())
  if whole:
    assert hashlib.sha256(chunk32.tobytes()).digest() == fix['steps_' + tag][t].tobytes(), 'frame %d (%s)' % (t, tag)


def oracle_frames(name, off, n, steps):
  t = helpers.load_template(name)
  t.param[0], t.param[1], t.param[2], t.param[3] = 0x5EED, 0, off & 0xFFFFFFFF, off >> 32  # RNG seed, global index of environment 0
  orc = binding.OracleEngine(t, n)
  orc.reset()
  for f in range(steps + 1):
    if f:
      orc.step_hashed(0x5EED, f - 1, 1, env_offset=off)
    planes = np.array(orc.planes)
    yield f, frame_digests(planes[:, 0], np.array(orc.reward), np.array(orc.reward_set), np.array(orc.discount), np.array(orc.done))


@pytest.mark.parametrize('name', GAMES)
def test_oracle_reproduces_a_sample_of_the_reference_digests(name):
  """CPU suite: the first 512 environments of the head and of the tail of the batch, all 256 steps (the fixtures say how
  many episodes ended on the way)."""
  fix = load(name)
  assert int(fix['n_envs'][0]) >= 4096 and int(fix['steps'][0]) >= 256
  if name != 'warehouse_L0':  # (uniform random actions do not finish a Sokoban level)
    assert int(fix['resets_head'][0]) > 0 and int(fix['resets_tail'][0]) > 0
  for tag in ('head', 'tail'):
    off = int(fix['offset_' + tag][0])
    for f, chunk32 in oracle_frames(name, off, 512, int(fix['steps'][0])):
      check_frame(fix, tag, f, chunk32, whole=False)


@pytest.mark.gpu
@pytest.mark.parametrize('name', GAMES)
def test_oracle_reproduces_every_reference_digest(name):
  fix = load(name)
  for tag in ('head', 'tail'):
    for f, chunk32 in oracle_frames(name, int(fix['offset_' + tag][0]), int(fix['n_envs'][0]), int(fix['steps'][0])):
      check_frame(fix, tag, f, chunk32)


def hip_frame(eng, lo, hi):
  planes = eng.planes_view()[lo:hi, 0].cpu().numpy()
  b = eng.buffers
  pick = lambda k: b[k].tensor[lo:hi].cpu().numpy()
  return frame_digests(planes, pick('reward'), pick('reward_set'), pick('discount'), pick('done'))


@pytest.mark.gpu
@pytest.mark.parametrize('name', GAMES)
def test_hip_reproduces_every_reference_digest(name):
  """4,096 environments x 256 steps at both offsets through Engine.step_hashed (env_offset): what the kernels return is
  what the imported reference returned, digest for digest."""
  from pycolab_amd.engine import Engine
  fix = load(name)
  t = helpers.load_template(name)
  n, steps = int(fix['n_envs'][0]), int(fix['steps'][0])
  for tag in ('head', 'tail'):
    off = int(fix['offset_' + tag][0])
    eng = Engine.from_template(t, batch=n, device=0, auto_reset=True, seed=0x5EED, env_offset=off)
    eng.its_showtime()
    for f in range(steps + 1):
      if f:
        eng.step_hashed(0x5EED, f - 1, 1, env_offset=off)
      check_frame(fix, tag, f, hip_frame(eng, 0, n))
    assert not eng.buffers['error'].tensor.any()
    eng.close()


@pytest.mark.gpu
def test_headline_engine_reproduces_the_reference_digests_at_its_head_and_tail():
  """The headline itself: ONE engine of 1,048,576 environments in its default launch shape, 256 steps; its first and its
  last 4,096 environments are the reference's, digest for digest."""
  from pycolab_amd import _native as N
  from pycolab_amd.engine import Engine
  fix = load('scrolly_maze_L0')
  t = helpers.load_template('scrolly_maze_L0')
  B, n, steps = 1048576, int(fix['n_envs'][0]), int(fix['steps'][0])
  assert int(fix['offset_tail'][0]) == B - n
  eng = Engine.from_template(t, batch=B, device=0, auto_reset=True, seed=0x5EED)
  eng.its_showtime()
  for f in range(steps + 1):
    if f:
      eng.step_hashed(0x5EED, f - 1, 1)
    check_frame(fix, 'head', f, hip_frame(eng, 0, n))
    check_frame(fix, 'tail', f, hip_frame(eng, B - n, B))
  assert int(N.lib().pcx_engine_launch_shape(eng._native)) in (3, 5)
  assert not eng.buffers['error'].tensor.any()
  eng.close()
