"""The oracle against the reference ITSELF on random games of Plot directives (CPU): walkers that pile up on a small
board, a static sprite and a drape, each with a random table of calls -- `change_z_order(move, in_front_of | None)`,
`add_reward(n)`, `terminate_episode([discount])`, several per update -- injected into the reference's own test entities
the way its `tests/engine_test.py:169-295` does (`tt.pre_update`; oracle/directive_scenarios.py `inject`) and carried
as data by this package's tabled prefabs, whose template the oracle steps.  Runtime z-order (engine.py:796-835) is
pinned by two recorded fixtures; this is the same comparison over tables nobody designed."""
import importlib
import sys
import warnings

import numpy as np
import pytest

from oracle import binding, directive_scenarios as ds, ref_live
from pycolab_amd.compiler import GameTemplate

pytestmark = pytest.mark.skipif(ref_live.reference_path() is None, reason='the reference is neither under /root/reference nor built under oracle/_ref')


def random_spec(rng):
  rows, cols = int(rng.randint(2, 5)), int(rng.randint(4, 9))
  art = np.full((rows, cols), '.', dtype='<U1')
  cells = [(r, c) for r in range(rows) for c in range(cols)]
  rng.shuffle(cells)
  walkers = 'abc'[:int(rng.randint(1, 4))]
  cast = list(walkers) + (['Q'] if rng.rand() < 0.5 else []) + ['D']
  for ch in cast[:-1]:
    art[cells.pop()] = ch
  for _ in range(int(rng.randint(1, 4))):  # the drape's curtain: a few cells, so that things overlap it
    art[cells.pop()] = 'D'

  def calls():
    table = {}
    for value in (1, 2, 3):
      if rng.rand() < 0.85:
        row = []
        for _ in range(int(rng.randint(1, 3))):
          kind = rng.rand()
          if kind < 0.6:
            move = cast[int(rng.randint(len(cast)))]
            others = [None] + [c for c in cast if c != move]
            row.append(('change_z_order', move, others[int(rng.randint(len(others)))]))
          elif kind < 0.85:
            row.append(('add_reward', int(rng.randint(-5, 12))))
          elif kind < 0.93:
            row.append(('terminate_episode',))
          else:
            row.append(('terminate_episode', float(rng.choice([0.0, 0.25, 0.5, 0.875]))))
        table[value] = row
    return table
  entities, shift = {}, 0
  for ch in cast:
    if ch in walkers:
      entities[ch] = dict(kind='walker', impassable='' if rng.rand() < 0.7 else 'D', confined=bool(rng.rand() < 0.7), motion=(shift, 15),
                          directive=(shift + 4, 3), calls=calls())
      shift += 6
    else:
      entities[ch] = dict(kind='sprite' if ch == 'Q' else 'drape', motion=None, directive=(shift, 3), calls=calls())
      shift += 2
  z = list(cast)
  rng.shuffle(z)
  order = list(cast)
  rng.shuffle(order)
  cut = int(rng.randint(1, len(order) + 1))
  return dict(art=[''.join(r) for r in art], beneath='.', z_order=''.join(z), schedule=[g for g in (order[:cut], order[cut:]) if g],
              entities=entities)


@pytest.mark.parametrize('seed', range(40))
def test_oracle_matches_the_live_reference_on_random_plot_directives(seed):
  path = ref_live.reference_path()
  if path not in sys.path:
    sys.path.insert(0, path)
  warnings.filterwarnings('ignore', category=DeprecationWarning)
  ref_art = importlib.import_module('pycolab.ascii_art')
  tt = importlib.import_module('pycolab.tests.test_things')
  from pycolab_amd import ascii_art
  from pycolab_amd.prefab_parts import tabled

  rng = np.random.RandomState(11000 + seed)
  spec = random_spec(rng)
  template = GameTemplate.from_engine(ds.build_twin(spec, ascii_art, tabled))
  E, T = 8, 120
  tape = np.stack([ds.tape(spec, rng, T) for _ in range(E)], axis=1)
  orc = binding.OracleEngine(template, E)
  orc.reset()
  now = lambda: (np.array(orc.planes)[:, 0].copy(), np.array(orc.reward), np.array(orc.reward_set), np.array(orc.discount), np.array(orc.done))
  frames = [now()]
  for t in range(T):
    orc.step(tape[t], auto_reset=True)
    assert not np.array(orc.error).any()
    frames.append(now())
  reordered = 0
  for e in range(E):
    make = lambda: ds.build_reference(spec, ref_art, tt)
    game = make()
    obs, r, d = game.its_showtime()
    for t in range(T + 1):
      if t:
        if game.game_over:
          game = make()
          obs, r, d = game.its_showtime()
        else:
          a = int(tape[t - 1, e])
          before = list(game.z_order)
          ds.inject(spec, game, a, tt)
          obs, r, d = game.play(ds.reference_action(spec, a))
          reordered += list(game.z_order) != before
      board, reward, reward_set, discount, done = frames[t]
      where = 'seed %d: env %d frame %d' % (seed, e, t)
      np.testing.assert_array_equal(obs.board, board[e], err_msg=where)
      assert (r is None) == (not reward_set[e]) and (r or 0) == reward[e], where
      assert d == discount[e] and game.game_over == bool(done[e]), where
  assert reordered > 0
