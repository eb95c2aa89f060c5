"""pcx_engine_export_state / _import_state (checkpoint/resume): an episode
resumed from a checkpoint in ANOTHER engine continues exactly as the engine that
made it -- every kernel family, RNG draw counters and sticky scroll state included."""
import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

CASES = [('scrolly_maze_L0', 700, 0), ('marauders', 500, 0), ('warehouse_L1', 600, 0), ('better_scrolly_maze_L1', 300, 0),
         ('hello_world', 400, 0), ('walkers_scroll_groups', 300, 0), ('directives_z_order', 300, 0), ('marauders', 300, 1)]


@pytest.mark.parametrize('name,batch,force_generic', CASES)
def test_resume_equals_uninterrupted(name, batch, force_generic, monkeypatch):
  from tests.hip_adapter import HipAdapter
  if force_generic:
    monkeypatch.setenv('PCX_FORCE_GENERIC', '1')
  t = helpers.load_template(name)
  a = HipAdapter(t, batch, seed=7)
  a.reset()
  a.step_hashed(0xC0FFEE, 0, 37)
  blob = a.eng.export_state(with_observation=True)
  assert blob.dtype == np.uint8 and blob.size > 64 * batch // 64
  b = HipAdapter(t, batch, seed=7)
  b.reset()
  b.step_hashed(0xBAD, 0, 5)              # somewhere else entirely
  b.eng.import_state(blob)
  for name_ in ('planes', 'reward', 'reward_set', 'discount', 'done', 'frame'):
    np.testing.assert_array_equal(b.read(name_), a.read(name_), err_msg='right after import: ' + name_)
  for chunk in range(3):
    auto = chunk != 1                      # one chunk leaves finished environments frozen (their frame must survive too)
    a.step_hashed(0xC0FFEE, 37 + 20 * chunk, 20, auto_reset=auto)
    b.step_hashed(0xC0FFEE, 37 + 20 * chunk, 20, auto_reset=auto)
    for name_ in ('planes', 'reward', 'reward_set', 'discount', 'done', 'frame', 'error'):
      np.testing.assert_array_equal(b.read(name_), a.read(name_), err_msg='chunk %d: %s' % (chunk, name_))
    np.testing.assert_array_equal(b.sprites(), a.sprites())
  other = HipAdapter(t, batch + 64, seed=7)
  other.reset()
  with pytest.raises(ValueError):          # a checkpoint of another batch is refused
    other.eng.import_state(blob)
