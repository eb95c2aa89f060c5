import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


# `PCX_FORCE_GENERIC=1 python -m pytest tests -m gpu` runs the whole GPU suite through pcx_generic_step (every shipped game
# the table-driven kernel can step: all but scrolly_maze).  These tests are about the hand-written kernels themselves --
# they name them, or use an epilogue kind / a launch shape / a fused window stack only those have -- and are skipped then.
HAND_WRITTEN_KERNELS_ONLY = {
    'test_which_kernel_steps_which_game', 'test_hip_hand_written_kernels_other_launch_shapes',
    'test_unoccluded_layers_from_the_hand_written_kernels', 'test_observation_planes_beyond_4_gib',
    'test_random_levels_match_oracle', 'test_random_marauders_layouts_match_oracle',
    'test_feature_array_fused_into_the_step_kernel', 'test_feature_array_fused_on_boards_of_any_size',
    'test_channels_last_feature_array_fused_into_the_step_kernel', 'test_channels_last_feature_array_in_the_scrolly_maze_mask_path',
    'test_value_array_fused_into_the_step_kernel', 'test_repainter_fused_into_the_step_kernel', 'test_epilogue_without_any_plane',
    'test_fused_epilogue_belongs_to_the_engine', 'test_fused_postprocessors_match_reference', 'test_fused_outputs_match_the_numpy_oracle',
    'test_fused_window_feature_stack_matches_reference', 'test_fused_window_feature_stack_refusals_and_big_batch',
    # round 5: the persistent workers of pcx_warehouse_step / pcx_hello_world_step (their launch shape is asserted)
    'test_stream_kernels_persistent_workers_match_oracle', 'test_stream_kernels_persistent_workers_equal_the_round_2_shape',
    'test_warehouse_persistent_workers_equal_the_round_2_shape_at_config_4', 'test_resume_with_a_feature_stack_fused_into_a_window',
}


def pytest_collection_modifyitems(config, items):
  if os.environ.get('PCX_FORCE_GENERIC') != '1':
    return
  skip = pytest.mark.skip(reason='PCX_FORCE_GENERIC=1: about a hand-written kernel (its name, launch shapes, epilogue kinds or window stack)')
  for item in items:
    if getattr(item, 'originalname', item.name) in HAND_WRITTEN_KERNELS_ONLY:
      item.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
  return GOLDEN
