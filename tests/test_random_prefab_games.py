"""The table-driven kernel (both of its builds) against the oracle on the random prefab-only games of
tests/test_reference_live_random_walkers.py -- rooms of MazeWalkers, scrolling worlds, one or two scrolling groups,
hidden walkers -- which the CPU suite holds the oracle to the live reference on: the error bits (the frame and kind of the first raise)
and, for as long as an environment has not raised, every output, every step, on uniform tapes (many environments raise) and on heading tapes (games run long)."""
import numpy as np
import pytest

from oracle import binding, walker_scenarios
from pycolab_amd.compiler import GameTemplate
from tests import helpers
from tests import test_reference_live_random_walkers as games
from tests.hip_adapter import HipAdapter

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('build', helpers.BUILDS)
@pytest.mark.parametrize('maker', [games.random_room, games.random_scroll, games.random_scroll2], ids=lambda m: m.__name__)
@pytest.mark.parametrize('seed', range(6))
def test_random_prefab_games_match_oracle(maker, seed, build, monkeypatch):
  from pycolab_amd import ascii_art
  from pycolab_amd.prefab_parts import tabled
  helpers.force_generic(monkeypatch, build)
  spec = maker(np.random.RandomState(9000 + seed))
  t = GameTemplate.from_engine(walker_scenarios.build(spec, ascii_art, tabled.TabledWalker, tabled.TabledScrolly, True))
  B, T, n_actions = 200, 64, int(t.n_actions)
  hip, orc = HipAdapter(t, B), binding.OracleEngine(t, B)
  hip.reset(); orc.reset()
  rng = np.random.RandomState(9700 + seed)
  headings = np.zeros((T, B), np.int32)
  for e in range(B):
    for f in range(max(1, spec['n_fields'])):
      headings[:, e] |= walker_scenarios.field_tape(rng, T, e % 2 == 0) << (4 * f)
  before = np.zeros(B, np.uint8)  # error bits so far: once the reference has raised, what an environment shows is nobody's law
  for step in range(T):
    a = rng.randint(0, n_actions, size=B).astype(np.int32) if step % 2 else headings[step]
    a[rng.rand(B) < 0.02] = -1
    hip.step(a, auto_reset=True); orc.step(a, auto_reset=True)
    err_h, err_o = hip.read('error'), np.array(orc.error)
    fresh = before == 0
    np.testing.assert_array_equal(err_h[fresh], err_o[fresh], err_msg='error bits after step %d' % (step + 1))  # same frame, same kind
    ok = err_o == 0
    for name in ('planes', 'reward', 'reward_set', 'discount', 'done', 'frame'):
      np.testing.assert_array_equal(hip.read(name)[ok], np.array(getattr(orc, name))[ok], err_msg='%s after step %d' % (name, step + 1))
    before |= err_o
  ok = before == 0
  np.testing.assert_array_equal(hip.sprites()[ok], orc.sprites()[ok])
  hip.eng.close()
