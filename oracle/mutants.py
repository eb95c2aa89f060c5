"""TEST INFRASTRUCTURE: mutants of the CPU oracle, to show that the fixtures have teeth.

The oracle (pcx_oracle.c, pcx_oracle_crop.c) is trusted because it reproduces what the reference recorded
(tests/golden/traces, tests/golden/reftests).  That trust is worth as much as the fixtures' power to tell a
correct restatement from a nearly correct one.  Each MUTANT below is one plausible mis-reading of the reference --
a quirk a restatement gets wrong when it follows intuition instead of the cited lines -- expressed as a textual
replacement in the oracle's source.  `build()` compiles the mutated text (the oracle's sources are never
changed), `loaded()` swaps the library in under oracle.binding, and tests/test_oracle_mutants.py requires the
fixtures named in `killed_by` to FAIL on it (and, for the mutant the shipped levels cannot tell apart,
`survives` to pass: DESIGN.md section 2).

  python -m oracle.mutants            prints the whole kill matrix (every mutant x every fixture; minutes)

Never imported by the product package.
"""
import contextlib
import ctypes
import gc
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ['pcx_oracle.c', 'pcx_oracle_crop.c']


class Mutant(object):

  def __init__(self, name, cite, source, old, new, killed_by, survives=()):
    self.name, self.cite, self.source, self.old, self.new = name, cite, source, old, new
    self.killed_by, self.survives = tuple(killed_by), tuple(survives)


# fixture ids: 'trace:<name>' (tests/golden/traces, test_oracle_golden), 'crop:<name>' (the croppers recorded with a
# trace, test_cropping), 'reftest:<name>' (the reference's own known-answer tests, tests/golden/reftests),
# 'engine_test:<what>' (tests/engine_test.py:169-295 restated in test_reference_known_answers), 'raise:<name>'
# (tests/golden/raises: where the reference raised, test_raise_parity)
MUTANTS = [
    Mutant('kill_test_on_true_positions', 'examples/scrolly_maze.py:304 compares VIRTUAL positions',
           'pcx_oracle.c',
           'if (s->vrow == P->vrow && s->vcol == P->vcol) plot_terminate(&env->plot, 0.0f); /* :304-305 */',
           'if (s->row == P->row && s->col == P->col) plot_terminate(&env->plot, 0.0f);',
           killed_by=['trace:scrolly_custom_F'],
           survives=['trace:scrolly_maze_L0', 'trace:scrolly_maze_L1', 'trace:scrolly_maze_L2']),
    Mutant('diagonal_without_the_flank_rule', 'prefab_parts/sprites.py:539-541: both flanks impassable block a diagonal',
           'pcx_oracle.c',
           'if (mw_blocked_at(e, d, s, board, dr, 0) && mw_blocked_at(e, d, s, board, 0, dc)) return 1;',
           '',
           killed_by=['reftest:testBasicWalking_0', 'trace:walkers_room']),
    Mutant('diagonal_blocked_by_either_flank', 'prefab_parts/sprites.py:539-541: BOTH flanks, not either',
           'pcx_oracle.c',
           'if (mw_blocked_at(e, d, s, board, dr, 0) && mw_blocked_at(e, d, s, board, 0, dc)) return 1;',
           'if (mw_blocked_at(e, d, s, board, dr, 0) || mw_blocked_at(e, d, s, board, 0, dc)) return 1;',
           killed_by=['reftest:testScrolly_0', 'trace:walkers_room']),
    Mutant('off_board_keeps_the_last_true_position', 'prefab_parts/sprites.py:391-411: off the board the true position is (0, 0)',
           'pcx_oracle.c',
           'else { s->row = 0; s->col = 0; }',
           'else { }',
           killed_by=['reftest:testNotConfinedToBoard_0', 'trace:scrolly_maze_L0']),
    Mutant('re_entry_always_visible', 'prefab_parts/sprites.py:223-275: re-entry restores the visibility SAVED at the exit',
           'pcx_oracle.c',
           'if (!old_on && new_on) s->visible = s->prior_visible;',
           'if (!old_on && new_on) s->visible = 1;',
           killed_by=['trace:walkers_hidden'],  # nothing else: the scenario was added because this mutant survived
           survives=['trace:walkers_room', 'trace:marauders', 'reftest:testNotConfinedToBoard_0']),
    Mutant('edge_blocks_everybody', 'prefab_parts/sprites.py:496-511: EDGE blocks only walkers confined to the board',
           'pcx_oracle.c',
           'return d->confined; /* EDGE */',
           'return 1;',
           killed_by=['reftest:testNotConfinedToBoard_0', 'trace:scrolly_maze_L0']),
    Mutant('prescroll_never_refreshed', 'prefab_parts/drapes.py:407-408: before the first _maybe_move of a frame, prescroll = current',
           'pcx_oracle.c',
           'if (s->last_maybe_move_frame < env->plot.frame) { /* :407-408 */',
           'if (0) {',
           killed_by=['trace:scrolly_maze_L0']),
    Mutant('margin_scroll_asks_about_the_order', 'prefab_parts/drapes.py:650-651: is_possible() is asked about the MOTION, not the order',
           'pcx_oracle.c',
           'can &= scroll_is_possible(p, dr, dc);',
           'can &= scroll_is_possible(p, o0, o1);',
           killed_by=['reftest:testScrolly_0', 'trace:walkers_scroll_margins']),
    Mutant('order_consistency_either_axis', 'prefab_parts/sprites.py:449-454: the egocentrist raises only if BOTH axes disagree',
           'pcx_oracle.c',
           'if (d->egocentric && order[0] != dr && order[1] != dc) env->error |= OX_ERR_SCROLL;',
           'if (d->egocentric && (order[0] != dr || order[1] != dc)) env->error |= OX_ERR_SCROLL;',
           killed_by=['reftest:testScrolly_1', 'trace:walkers_scroll_groups']),
    Mutant('judge_rewards_the_count', 'examples/warehouse_manager.py:260: the reward is the CHANGE of boxes on goals',
           'pcx_oracle.c',
           'plot_add_reward(&env->plot, on_goals - d->var[0]); /* :260 */',
           'plot_add_reward(&env->plot, on_goals);',
           killed_by=['trace:warehouse_L0']),
    Mutant('marauder_period_plain_division', 'examples/extraterrestrial_marauders.py:157: total // 8.0000001',
           'pcx_oracle.c',
           'int period = (total - 1) / 8;',
           'int period = total / 8;',
           killed_by=['trace:marauders']),
    Mutant('every_bolt_fires', 'examples/extraterrestrial_marauders.py:213-217: one shot per frame',
           'pcx_oracle.c',
           'if (env->plot.kv[EM_LAST_PLAYER_SHOT] == env->plot.frame) return;',
           '',
           killed_by=['trace:marauders']),
    Mutant('one_repaint_per_step', 'engine.py:726-735: a repaint after EVERY update group',
           'pcx_oracle.c',
           'render(e, b); /* :735 */',
           'if (g == e->t.n_groups - 1) render(e, b);',
           killed_by=['trace:warehouse_L0']),
    Mutant('reward_zero_instead_of_none', 'engine.py:761-790, plot.py:200-226: no add_reward -> reward None',
           'pcx_oracle.c',
           'e->reward_set[b] = (uint8_t)env->plot.reward_set;',
           'e->reward_set[b] = 1;',
           killed_by=['trace:scrolly_maze_L0']),
    Mutant('z_order_change_without_repaint', 'engine.py:632-637: a z-order change repaints before the observation is returned',
           'pcx_oracle.c',
           'if (env->plot.n_z_updates) render(e, b);',
           'if (0) render(e, b);',
           killed_by=['engine_test:z_order', 'trace:directives_z_order']),
    Mutant('moved_thing_goes_behind_its_anchor', 'engine.py:796-835: change_z_order(move, in_front_of)',
           'pcx_oracle.c',
           '      order[n++] = id;\n      if (id == front) order[n++] = move;',
           '      if (id == front) order[n++] = move;\n      order[n++] = id;',
           killed_by=['engine_test:z_order']),
    Mutant('crop_overhang_never_raises', 'cropping.py:175-183: without pad_char an overhang raises',
           'pcx_oracle_crop.c',
           'if (top < 0 || left < 0 || bottom > R || right > C) { c->error[b] = 1; return; } /* :175-183 */',
           '',
           killed_by=['raise:fixed_crop_overhang']),  # nothing else: its CPU test was added because this mutant survived
    Mutant('drape_centroid_is_the_mean', 'cropping.py:598: per-axis MEDIAN of the curtain\'s cells',
           'pcx_oracle.c',
           '  *row = median_int(rs, m);\n  *col = median_int(cs, m);',
           '  { long a = 0, c2 = 0; for (int i = 0; i < m; ++i) { a += rs[i]; c2 += cs[i]; } *row = (int)(a / m); *col = (int)(c2 / m); }',
           killed_by=['crop:marauders', 'crop:warehouse_L1']),
    Mutant('pan_without_the_edge_exception', 'cropping.py:491-504: at the board\'s edge the window lets the centroid into the margin',
           'pcx_oracle_crop.c',
           'if (c->d.pad_char < 0) { /* :491-504 */',
           'if (0) {',
           killed_by=['crop:scrolly_maze_L0', 'crop:better_scrolly_maze_L1']),
    Mutant('saccade_keeps_the_initial_offset', 'cropping.py:414-415: a saccade recentres WITHOUT initial_offset',
           'pcx_oracle_crop.c',
           'initialise(c, corner, 1, crow, ccol, rows / 2, cols / 2);',
           'initialise(c, corner, 1, crow, ccol, rows / 2 + c->d.initial_offset_rows, cols / 2 + c->d.initial_offset_cols);',
           killed_by=['reftest:testScrollingInitialOffset_0', 'crop:better_scrolly_maze_L1']),
]


def mutated_sources(mutant):
  out = {}
  for name in SOURCES:
    text = open(os.path.join(HERE, name)).read()
    if name == mutant.source:
      if text.count(mutant.old) != 1:
        raise ValueError('mutant %s: its anchor occurs %d times in %s (the oracle moved on: update oracle/mutants.py)'
                         % (mutant.name, text.count(mutant.old), name))
      text = text.replace(mutant.old, mutant.new)
    out[name] = text
  return out


def build(mutant, out_dir):
  """Compiles the mutated text into out_dir/liboracle_<name>.so and returns the path."""
  srcs = []
  for name, text in mutated_sources(mutant).items():
    p = os.path.join(out_dir, mutant.name + '_' + name)
    with open(p, 'w') as f:
      f.write(text)
    srcs.append(p)
  so = os.path.join(out_dir, 'liboracle_%s.so' % mutant.name)
  subprocess.check_call([os.environ.get('CC', 'gcc'), '-O1', '-fPIC', '-std=c11', '-w', '-shared', '-I', HERE,
                         '-I', os.path.join(HERE, '..', 'include'), '-o', so] + srcs)
  return so


@contextlib.contextmanager
def loaded(so):
  """Everything that goes through oracle.binding runs on the library `so` inside the block."""
  from oracle import binding
  from pycolab_amd import _native as N
  real = binding.lib()
  gc.collect()
  binding._lib = N.bind(ctypes.CDLL(so), binding._SYMS)
  try:
    yield
  finally:
    gc.collect()  # engines of the block go through the library that made them
    binding._lib = real


def fixture_passes(fixture):
  """Runs the suite's own check of one fixture against whatever library oracle.binding holds."""
  kind, name = fixture.split(':', 1)
  try:
    if kind == 'trace':
      from tests import test_oracle_golden
      test_oracle_golden.test_oracle_matches_reference_trace(name)
    elif kind == 'crop':
      from tests import test_cropping
      test_cropping.test_oracle_croppers_match_reference(name)
    elif kind == 'reftest':
      from tests import test_reference_known_answers
      test_reference_known_answers.test_oracle_reproduces_reference_known_answers(name)
    elif kind == 'engine_test':
      from tests import test_reference_known_answers as t
      if name == 'z_order':
        t.test_oracle_z_order_change_known_answer()
      else:
        for discount in (None, 0.5):
          t.test_oracle_reward_and_episode_end_known_answer(discount)
    elif kind == 'raise':
      from tests import test_raise_parity
      if name == 'fixed_crop_overhang':
        test_raise_parity.test_oracle_fixed_cropper_without_pad_raises_where_the_reference_does()
      else:
        test_raise_parity.test_oracle_error_bit_rises_where_the_reference_raised(name)
    else:
      raise ValueError(fixture)
  except AssertionError:
    return False
  return True


def all_fixtures():
  from tests import test_oracle_golden, test_cropping, test_reference_known_answers, test_raise_parity
  return (['trace:' + n for n in test_oracle_golden.ALL_TRACES] + ['crop:' + n for n in test_cropping.CROPPED] +
          ['reftest:' + n for n in test_reference_known_answers.NAMES] + ['engine_test:z_order', 'engine_test:reward'] +
          ['raise:' + n for n in test_raise_parity.WALKERS + ('fixed_crop_overhang',)])


def main():
  sys.path.insert(0, os.path.join(HERE, '..'))
  fixtures = all_fixtures()
  with tempfile.TemporaryDirectory(prefix='pcx_mutants_') as tmp:
    for m in MUTANTS:
      with loaded(build(m, tmp)):
        killers = [f for f in fixtures if not fixture_passes(f)]
      print('%-42s killed by %d of %d: %s' % (m.name, len(killers), len(fixtures), ' '.join(killers) or '-- SURVIVES --'))
      sys.stdout.flush()


if __name__ == '__main__':
  main()
