"""Pins the CPU oracle (oracle/pcx_oracle.c) against golden traces recorded
from the imported reference itself (oracle/gen_golden.py)."""
import numpy as np
import pytest

from oracle import binding
from tests import helpers


class OracleAdapter(binding.OracleEngine):

  def read(self, name):
    return np.array(getattr(self, name))


ALL_TRACES = ['scrolly_maze_L0', 'scrolly_maze_L1', 'scrolly_maze_L2', 'warehouse_L0', 'warehouse_L1',
              'warehouse_L2', 'hello_world', 'marauders', 'scrolly_maze_L1_unoccluded',
              'warehouse_L0_unoccluded', 'marauders_unoccluded', 'walkers_room', 'walkers_hidden',
              'walkers_scroll_margins', 'walkers_scroll_always', 'walkers_scroll_groups', 'better_scrolly_maze_L0',
              'better_scrolly_maze_L1', 'better_scrolly_maze_L2',
              # unshipped scrolly_maze levels (oracle/custom_levels.py): other board shapes, sprite sets, z-orders
              'scrolly_custom_A', 'scrolly_custom_B', 'scrolly_custom_C', 'scrolly_custom_D', 'scrolly_custom_E',
                      'scrolly_custom_A_unoccluded', 'scrolly_custom_C_unoccluded', 'scrolly_custom_E_unoccluded',
                      'scrolly_custom_F', 'scrolly_custom_G', 'scrolly_custom_H', 'warehouse_custom_A', 'warehouse_custom_B', 'marauders_custom_A', 'hello_custom_A',
              # shapes the hand-written kernels take with their run-time-shape instances
              'warehouse_custom_C', 'warehouse_custom_D', 'better_scrolly_custom_A', 'better_scrolly_custom_B', 'better_scrolly_custom_C', 'better_scrolly_custom_D', 'better_scrolly_custom_E',
              # Plot directives from inside update(): add_reward, terminate_episode(discount), change_z_order
              # (oracle/directive_scenarios.py; reference: tests/engine_test.py:169-295)
              'directives_z_order', 'directives_reward_discount', 'directives_two_discounts', 'directives_float_rewards']


@pytest.mark.parametrize('name', ALL_TRACES)
def test_oracle_matches_reference_trace(name):
  trace = helpers.load_trace(name)
  helpers.replay_trace(OracleAdapter, trace)


def test_action_hash_twins_agree():
  envs = np.arange(0, 5000, 37, dtype=np.uint64)
  for t in (0, 1, 999, 2**33):
    got = binding.action_hash_np(0x5EED, envs, np.uint64(t))
    want = [binding.action_hash(0x5EED, int(e), t) for e in envs]
    np.testing.assert_array_equal(got, np.array(want, np.uint32))
