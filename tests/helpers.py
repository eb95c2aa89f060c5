"""Shared helpers for the parity tests."""
import os

import numpy as np

from pycolab_amd.compiler import GameTemplate

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
GOLDEN_RNG_SEED = 0x5EED


def to_np(x):
  """NumPy copy of a NumPy array or a (device) tensor."""
  return x.detach().cpu().numpy() if hasattr(x, 'detach') else np.asarray(x)


def load_template(name):
  return GameTemplate.load(os.path.join(GOLDEN, 'templates', name + '.npz'))


def load_trace(name):
  z = np.load(os.path.join(GOLDEN, 'traces', name + '.npz'))
  tr = {k: z[k] for k in z.files}
  tr['template'] = bytes(tr['template']).decode()
  return tr


def load_trace_raw(name):
  z = np.load(os.path.join(GOLDEN, 'traces', name + '.npz'))
  return {k: z[k] for k in z.files}


def expected_planes(boards, chars):
  """[E, R, C] boards -> [E, 1+L, R, C] planes with occluded layers
  (rendering.py:177-179: layer[c] = board == ord(c))."""
  layers = [(boards == c).astype(np.uint8) for c in chars]
  return np.stack([boards] + layers, axis=1)


def replay_trace(engine_factory, trace):
  """Drive an engine (oracle or HIP wrapper exposing reset/step/arrays) over a
  golden trace and compare every step.  `engine_factory(template, batch)` must
  return an object with reset(), step(actions, auto_reset), and numpy-able
  attributes planes/reward/reward_set/discount/done (+ sprites())."""
  template = load_template(trace['template'])
  template.param[0] = GOLDEN_RNG_SEED  # marauders: oracle/gen_golden.py ChoicePatch(seed=...)
  T, E = trace['actions'].shape
  eng = engine_factory(template, E)
  eng.reset()
  chars = list(trace['chars'])
  assert bytes(chars) == template.chars

  def check(t):
    where = 'frame index %d' % t
    if 'layers' in trace:  # occlusion_in_layers=False: layers were recorded
      want = np.concatenate([trace['boards'][t][:, None], trace['layers'][t]], axis=1)
    else:
      want = expected_planes(trace['boards'][t], chars)
    np.testing.assert_array_equal(eng.read('planes'), want, err_msg=where)
    np.testing.assert_array_equal(eng.read('reward_set'), trace['reward_set'][t], err_msg=where)
    np.testing.assert_array_equal(eng.read('reward'), trace['reward'][t], err_msg=where)
    np.testing.assert_array_equal(eng.read('discount'), trace['discount'][t], err_msg=where)
    np.testing.assert_array_equal(eng.read('done'), trace['done'][t], err_msg=where)
    assert not eng.read('error').any(), where
    np.testing.assert_array_equal(eng.sprites(), trace['sprites'][t], err_msg=where)

  check(0)
  for t in range(T):
    eng.step(trace['actions'][t], auto_reset=True)
    check(t + 1)
  return eng


_JIT_DIR = []


def force_generic(monkeypatch, build='table-driven'):
  """Step every game of this test with pcx_generic_step -- its table-driven build, or (build='specialised') the build
  hiprtc makes for the engine's template, which only engines of 4,096 environments and more get by themselves.  The
  code objects of a test process share one scratch cache directory."""
  import tempfile
  if not _JIT_DIR:
    _JIT_DIR.append(tempfile.mkdtemp(prefix='pcx_jit_'))
  monkeypatch.setenv('PCX_FORCE_GENERIC', '1')
  monkeypatch.setenv('PCX_GENERIC_JIT', '1' if build == 'specialised' else '0')
  monkeypatch.setenv('PCX_JIT_CACHE', _JIT_DIR[0])


BUILDS = ['table-driven', 'specialised']
