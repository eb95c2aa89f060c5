"""pcx_generic_step_pw (pcx_generic_kernel.h, round 6): the table-driven kernel as persistent workgroups whose first waves
only step work units (logic workers, two hand-over slots each) and whose other waves only stream them (render workers).
Both builds of it -- the interpreter in libpcx.so and the one hiprtc makes for the template -- against the oracle, step by
step, in shapes that make a worker walk many units (few workgroups, tickets with stealing), leave logic workers without a
unit, give one render worker several logic workers, and with environments left alone (their planes must survive) and
episodes restarted by the kernel.  The launch shape is opt-in (PCX_GENERIC_PW=1; launch shapes 32 / 33): measured, it ties
with pcx_generic_step's one-workgroup-per-group shape on warehouse_L0 and loses on games whose per-lane arrays are smaller
(profiles/r06_generic.md) -- engines of test size take it because PCX_GENERIC_PW_MIN=0 says so."""
import numpy as np
import pytest

from pycolab_amd import _native as N
from tests import helpers
from tests.test_generic_specialised import engines, same, shape_of

pytestmark = pytest.mark.gpu

GAMES = ['warehouse_L0', 'warehouse_custom_B', 'warehouse_L0_unoccluded', 'marauders', 'marauders_custom_A', 'marauders_unoccluded',
         'walkers_room', 'walkers_hidden', 'walkers_scroll_groups', 'walkers_scroll_margins', 'directives_z_order',
         'directives_reward_discount', 'hello_world', 'better_scrolly_maze_L1', 'better_scrolly_custom_B']
# logic workers x render workers, workgroups in the grid (0: as many as the batch wants), tickets
SHAPES = {'6x2': (6, 2, 0, 0), '3x1 two workgroups, tickets': (3, 1, 2, 1), '1x1 one workgroup': (1, 1, 1, 0), '5x3 tickets': (5, 3, 2, 1),
          '8x2 more workers than units': (8, 2, 0, 0)}


def knobs(monkeypatch, shape):
  nl, nr, grid, dynamic = SHAPES[shape]
  monkeypatch.setenv('PCX_GENERIC_PW', '1')
  monkeypatch.setenv('PCX_GENERIC_PW_MIN', '0')
  monkeypatch.setenv('PCX_GENERIC_PW_LOGIC', str(nl))
  monkeypatch.setenv('PCX_GENERIC_PW_RENDER', str(nr))
  if grid:
    monkeypatch.setenv('PCX_GENERIC_PW_GRID', str(grid))
  monkeypatch.setenv('PCX_GENERIC_PW_DYNAMIC', str(dynamic))


# every other game in the plain shape (round 6: the shape is opt-in; the games left out are stepped by the other shapes below),
# and every other shape on three of the games (the shapes differ in scheduling, which no game's
# rules touch: fifteen games x five shapes re-proved the same code 75 times and cost the GPU suite two minutes)
CASES = [(name, '6x2') for name in GAMES[::2] + ['marauders']] + [(GAMES[(3 * i + j) % len(GAMES)], shape) for i, shape in enumerate(s for s in sorted(SHAPES) if s != '6x2')
                                             for j in range(3)]


@pytest.mark.parametrize('name,shape', CASES)
def test_persistent_workers_match_oracle(name, shape, monkeypatch, tmp_path):
  knobs(monkeypatch, shape)
  B, T = (64 * 4 + 21 if 'more workers' in shape else 64 * 17 + 5), 30  # (round 6: the shape is opt-in; 17 units keep every worker walking several)
  t, spec, table, orc = engines(name, B, monkeypatch, tmp_path)
  same(spec, orc, 'frame 0')
  rng = np.random.RandomState(11)
  na = max(1, int(t.n_actions))
  for step in range(T):
    if step % 7 == 3:  # the shared hashed tape, several launches
      spec.step_hashed(0x5EED, step, 2); table.step_hashed(0x5EED, step, 2); orc.step_hashed(0x5EED, step, 2)
    else:
      a = rng.randint(0, na, size=B).astype(np.int32)
      r = rng.rand(B)
      a[r < 0.04] = -1
      a[(r >= 0.04) & (r < 0.06)] = na + rng.randint(0, 30)
      auto = step % 5 != 0  # (every fifth step leaves finished environments alone: their observation must stay)
      spec.step(a, auto_reset=auto); table.step(a, auto_reset=auto); orc.step(a, auto_reset=auto)
    # (occlusion_in_layers=False: layers are raw masks, not board == character -- the owner-code hand-over cannot say that, and
    # the launch stays with pcx_generic_step)
    assert (shape_of(spec), shape_of(table)) == ((31, 30) if name.endswith('_unoccluded') else (33, 32))
    same(spec, orc, '%s, specialised build, step %d' % (shape, step))
    if step % 4 == 3:
      same(table, orc, '%s, table-driven build, step %d' % (shape, step))
  same(table, orc, '%s, table-driven build, the end' % shape)
  assert N.lib().pcx_engine_kernel_name(spec.eng._native).decode() == 'pcx_generic_step'
  spec.eng.close(); table.eng.close()


@pytest.mark.parametrize('name', ['warehouse_L0', 'walkers_scroll_groups', 'marauders_custom_A'])
def test_persistent_workers_with_their_tuner_equal_the_one_group_shape_at_eight_units_per_cu(name, monkeypatch, tmp_path):
  """131,072 environments (eight units per CU), PCX_GENERIC_PW=1 and no other knob: the launch takes the persistent workers
  (its tuner's candidates take turns on the first launches) and writes, over the whole batch, what pcx_generic_step's
  one-workgroup-per-group shape writes."""
  import torch
  B, T = 131072, 44
  t, spec, _, _ = engines(name, 64, monkeypatch, tmp_path)  # (compiles the build; the small engines are not used)
  from tests.hip_adapter import HipAdapter
  monkeypatch.setenv('PCX_FORCE_GENERIC', '1')
  monkeypatch.setenv('PCX_JIT_CACHE', str(tmp_path))
  monkeypatch.setenv('PCX_GENERIC_PW', '1')
  pw = HipAdapter(t, B); pw.reset()
  monkeypatch.setenv('PCX_GENERIC_PW', '0')
  ref = HipAdapter(t, B); ref.reset()
  monkeypatch.setenv('PCX_GENERIC_PW', '1')
  for t0 in range(T):
    pw.step_hashed(0xFACE, t0, 1)
    assert shape_of(pw) == 33
    if t0 == T - 2:
      torch.cuda.synchronize()  # (the tuner settles at the first launch AFTER its last measuring launch has completed)
  monkeypatch.setenv('PCX_GENERIC_PW', '0')
  ref.step_hashed(0xFACE, 0, T)
  assert shape_of(ref) == 31
  assert torch.equal(pw.eng.planes_view(), ref.eng.planes_view())
  for nm in ('reward', 'reward_set', 'discount', 'done', 'frame', 'error'):
    assert torch.equal(pw.eng.buffers[nm].tensor, ref.eng.buffers[nm].tensor), nm
  assert pw.eng.tuner_done()
  pw.eng.close(); ref.eng.close()
