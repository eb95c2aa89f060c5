"""The table-driven kernel (both of its builds) against the oracle on the random Plot-directive games of
tests/test_reference_live_random_directives.py (runtime z-order changes, rewards, episode ends with discounts, several
per update), which the CPU suite holds the oracle to the live reference on: every output, every step."""
import numpy as np
import pytest

from oracle import binding, directive_scenarios as ds
from pycolab_amd.compiler import GameTemplate
from tests import helpers
from tests import test_reference_live_random_directives as games
from tests.hip_adapter import HipAdapter

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('build', helpers.BUILDS)
@pytest.mark.parametrize('seed', range(8))
def test_random_directive_games_match_oracle(seed, build, monkeypatch):
  from pycolab_amd import ascii_art
  from pycolab_amd.prefab_parts import tabled
  helpers.force_generic(monkeypatch, build)
  rng = np.random.RandomState(11000 + seed)
  spec = games.random_spec(rng)
  t = GameTemplate.from_engine(ds.build_twin(spec, ascii_art, tabled))
  B, T = 160, 72
  tape = np.stack([ds.tape(spec, rng, T) for _ in range(B)], axis=1)
  hip, orc = HipAdapter(t, B), binding.OracleEngine(t, B)
  hip.reset(); orc.reset()
  for step in range(T):
    auto = step % 5 != 2  # (every fifth step leaves finished environments finished)
    hip.step(tape[step], auto_reset=auto); orc.step(tape[step], auto_reset=auto)
    for name in ('planes', 'reward', 'reward_set', 'discount', 'done', 'frame', 'error'):
      np.testing.assert_array_equal(hip.read(name), np.array(getattr(orc, name)), err_msg='%s after step %d' % (name, step + 1))
  np.testing.assert_array_equal(hip.sprites(), orc.sprites())
  hip.eng.close()
