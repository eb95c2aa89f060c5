"""GPU: uses of the host API around the step path that the parity tests do
not reach -- engines on other HIP streams, interleaved engines, create/close
cycles, calls after close()."""
import numpy as np
import pytest

from oracle import binding
from pycolab_amd import cropping
from tests import helpers

pytestmark = pytest.mark.gpu


def _oracle_planes(t, batch, tape):
  orc = binding.OracleEngine(t, batch)
  orc.reset()
  for a in tape:
    orc.step(a, auto_reset=True)
  return np.array(orc.planes), np.array(orc.reward)


def test_engines_on_their_own_streams_interleaved():
  """Two engines of different games, each stepped on its own (non-default)
  stream, launches interleaved from one host thread; a third on the default
  stream in between.  Everything matches the oracle."""
  import torch
  from pycolab_amd.engine import Engine
  names = ['scrolly_maze_L0', 'warehouse_L0', 'marauders']
  ts = [helpers.load_template(n) for n in names]
  for t in ts:
    t.param[0] = 77
  B, T = 700, 40
  rng = np.random.RandomState(5)
  tapes = [rng.randint(0, int(t.n_actions), size=(T, B)).astype(np.int32) for t in ts]
  streams = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.current_stream()]
  engs = []
  for t, s in zip(ts, streams):
    with torch.cuda.stream(s):
      e = Engine.from_template(t, batch=B, auto_reset=True, seed=77)
      e.its_showtime()
      engs.append(e)
  dev_tapes = []
  for tape, s in zip(tapes, streams):
    with torch.cuda.stream(s):
      dev_tapes.append(torch.from_numpy(tape).cuda())
  torch.cuda.synchronize()
  for step in range(T):
    for e, tape, s in zip(engs, dev_tapes, streams):
      with torch.cuda.stream(s):
        e.step(tape[step])
  torch.cuda.synchronize()
  for e, t, tape, name in zip(engs, ts, tapes, names):
    want_planes, want_reward = _oracle_planes(t, B, tape)
    np.testing.assert_array_equal(e.planes_view(host=True), want_planes, err_msg=name)
    np.testing.assert_array_equal(e.buffers['reward'].numpy(), want_reward, err_msg=name)
    e.close()


def test_create_close_cycles_release_device_memory():
  import torch
  from pycolab_amd.engine import Engine
  t = helpers.load_template('better_scrolly_maze_L0')
  free0 = None
  for i in range(12):
    eng = Engine.from_template(t, batch=4096, auto_reset=True)
    cr = cropping.ScrollingCropper(10, 30, ['P'], initial_offset=(-2, -12))
    cr.set_engine(eng)
    cropping.fuse_croppers(eng, [cr])
    obs = eng.its_showtime()[0]
    eng.play(np.zeros(4096, np.int32))
    assert cr.crop(obs).board.shape == (4096, 10, 30)
    eng.close()
    del eng, cr, obs
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if i == 2:
      free0 = free
    if i > 2:  # (the first cycles warm allocator pools and code objects)
      assert free >= free0 - (8 << 20), 'device memory shrinks with every create/close cycle: %d -> %d' % (free0, free)


def test_calls_after_close_raise():
  from pycolab_amd.engine import Engine
  t = helpers.load_template('warehouse_L0')
  eng = Engine.from_template(t, batch=16, auto_reset=True)
  cr = cropping.FixedCropper((0, 0), 4, 4)
  cr.set_engine(eng)
  obs = eng.its_showtime()[0]
  cr.crop(obs)
  eng.close()
  eng.close()  # idempotent
  with pytest.raises((RuntimeError, ValueError)):
    eng.play(np.zeros(16, np.int32))
  with pytest.raises((RuntimeError, ValueError)):
    cr.crop(obs)
