#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: tests/golden/raises/*.npz -- tapes on which the REFERENCE RAISES, with the frame and the
exception type it raised at, per environment (VERDICT r4 #4 ii: the error bit used to be compared HIP <-> oracle only).

The batched engine cannot raise for one environment of a million: it sets that environment's error bit (sticky within the
episode) and goes on, and check_errors() / the asynchronous polls raise on the host.  These fixtures pin WHEN the bit must
come up and that everything returned before that frame is the reference's:

  walkers_*            the prefab scenarios of oracle/walker_scenarios.py under UNIFORM actions (the golden traces steer
                       away from this): scrolling orders with no component in common with an egocentric walker's motion
                       -> RuntimeError, prefab_parts/sprites.py:449-454; impossible / second orders -> scrolling.Error,
                       protocols/scrolling.py:412,519,524
  marauders_to_array   extraterrestrial_marauders with an ObservationToArray whose value mapping lacks the player's third
                       bolt: RuntimeError the first frame that bolt is on the board, rendering.py:517-522
  fixed_crop_overhang  FixedCropper without a pad character whose window leaves the board: RuntimeError at every crop,
                       cropping.py:175-183

Actions are the hashed tape (pcx_action_hash(seed, env, t) % n_actions), so the code under test replays them with
step_hashed.  Needs /root/reference (the reference's test entities live in pycolab/tests/test_things.py).
"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('PCX_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
warnings.filterwarnings('ignore', category=DeprecationWarning)

from oracle import binding, ref_live, walker_scenarios  # noqa: E402

SEED = 0x5EED
# -> pcx error bits (pcx_device.h ERR_INDEX / ERR_SCROLL).  ValueError: a Scrolly ordered beyond its own pattern --
# np.copyto cannot broadcast the clamped slice (drapes.py:689-695); the engine files it with the index errors.
KINDS = {'IndexError': 1, 'ValueError': 1, 'Error': 2, 'RuntimeError': 2}


def out_path(name):
  root = os.environ.get('PCX_GOLDEN_OUT') or os.path.join(ROOT, 'tests', 'golden')
  os.makedirs(os.path.join(root, 'raises'), exist_ok=True)
  return os.path.join(root, 'raises', name + '.npz')


def walkers(name, E, T):
  from pycolab import ascii_art as ref_art
  from pycolab.tests import test_things as tt
  from pycolab_amd.compiler import GameTemplate
  spec = walker_scenarios.SCENARIOS.get(name) or walker_scenarios.RAISING_SCENARIOS[name]
  n_actions = GameTemplate.load(os.path.join(ROOT, 'tests', 'golden', 'templates', name + '.npz')).n_actions
  names = walker_scenarios.MOTION_NAMES
  if spec['kind'] == 'scroll2':
    fields = {ch: w['field'] for w in spec['worlds'] for ch in list(w['scrollies']) + list(w['walkers'])}
    ref_action = lambda a: {ch: names[min((a >> sh) & mk, 8)] for ch, (sh, mk) in fields.items()}
  elif spec['n_fields']:
    fields = {ch: spec['walkers'][ch]['field'] for ch in spec['walkers']}
    ref_action = lambda a: {ch: names[min((a >> sh) & mk, 8)] for ch, (sh, mk) in fields.items()}
  else:
    ref_action = lambda a: names[min(a, 8)]
  boards, raise_frame, raise_bit, raise_type = None, np.full(E, -1, np.int32), np.zeros(E, np.uint8), []
  for e in range(E):
    game = walker_scenarios.build(spec, ref_art, tt.TestMazeWalker, tt.TestScrolly, False)
    obs, _, _ = game.its_showtime()
    if boards is None:
      boards = np.zeros((T + 1, E) + obs.board.shape, np.uint8)
    boards[0, e] = obs.board
    kind = ''
    for t in range(T):
      a = int(binding.action_hash(SEED, e, t) % n_actions)
      try:
        obs, _, _ = game.play(ref_action(a))
      except Exception as ex:  # pylint: disable=broad-except
        kind = type(ex).__name__
        raise_frame[e], raise_bit[e] = t + 1, KINDS[kind]
        break
      assert not game.game_over
      boards[t + 1, e] = obs.board
    raise_type.append(kind)
  np.savez_compressed(out_path(name), template=np.frombuffer(name.encode(), np.uint8), seed=np.array([SEED], np.uint64),
                      boards=boards, raise_frame=raise_frame, raise_bit=raise_bit,
                      raise_type=np.frombuffer('\n'.join(raise_type).encode(), np.uint8))
  print('%s: %d of %d environments raised within %d steps (%s)' % (name, int((raise_frame >= 0).sum()), E, T,
                                                                     sorted(set(k for k in raise_type if k))), flush=True)


def unwalled(name, make, E, T, count=None):
  """A level without walls around it (oracle/custom_levels.py) on the hashed tape, episodes restarted the way the
  batched engines restart them (a finished environment is rebuilt at the next step), until the reference raises.
  `count(game, action, board)`: statistics of the events the fixture is there for (printed, not stored)."""
  from pycolab_amd.compiler import GameTemplate
  n_actions = GameTemplate.load(os.path.join(ROOT, 'tests', 'golden', 'templates', name + '.npz')).n_actions
  boards, raise_frame, raise_bit, raise_type = None, np.full(E, -1, np.int32), np.zeros(E, np.uint8), []
  episodes = events = 0
  for e in range(E):
    game = make()
    obs, _, _ = game.its_showtime()
    if boards is None:
      boards = np.zeros((T + 1, E) + obs.board.shape, np.uint8)
    boards[0, e] = obs.board
    kind = ''
    for t in range(T):
      try:
        if game.game_over:
          game = make()
          obs, _, _ = game.its_showtime()
          episodes += 1
        else:
          a = int(binding.action_hash(SEED, e, t) % n_actions)
          if count:
            events += count(game, a, obs.board)
          obs, _, _ = game.play(a)
      except Exception as ex:  # pylint: disable=broad-except
        kind = type(ex).__name__
        raise_frame[e], raise_bit[e] = t + 1, KINDS[kind]
        break
      boards[t + 1, e] = obs.board
    raise_type.append(kind)
  np.savez_compressed(out_path(name), template=np.frombuffer(name.encode(), np.uint8), seed=np.array([SEED], np.uint64),
                      boards=boards, raise_frame=raise_frame, raise_bit=raise_bit,
                      raise_type=np.frombuffer('\n'.join(raise_type).encode(), np.uint8))
  print('%s: %d of %d environments raised within %d steps (%s); %d episodes restarted, %d of the events it is there for'
        % (name, int((raise_frame >= 0).sum()), E, T, sorted(set(k for k in raise_type if k)), episodes, events), flush=True)


def warehouse_open(name, E, T):
  from pycolab import ascii_art as ref_art
  from pycolab.examples import warehouse_manager
  from oracle import custom_levels

  def pushes_through_index_minus_one(game, a, board):
    n, P = 0, game.things['P']
    for c, box in game.things.items():
      if c.isdigit() and box.visible and P.visible:
        r, col = box.position
        n += ((a == 1 and r == 0 and tuple(P.position) == (board.shape[0] - 1, col)) or
              (a == 3 and col == 0 and tuple(P.position) == (r, board.shape[1] - 1)))
    return n
  unwalled(name, lambda: custom_levels.make_warehouse(name, warehouse_manager, ref_art), E, T, pushes_through_index_minus_one)


def marauders_to_array(E, T):
  make, n_actions = ref_live._import_game('marauders')
  from pycolab import rendering
  choice = ref_live._Choice(SEED, binding.action_hash)
  real_choice, np.random.choice = np.random.choice, choice
  try:
    _marauders_to_array(E, T, make, n_actions, rendering, choice)
  finally:
    np.random.choice = real_choice  # 


def _marauders_to_array(E, T, make, n_actions, rendering, choice):
  mapping = {c: float(i) for i, c in enumerate(' BPXabdyz')}  # no 'c': the player's third bolt (extraterrestrial_marauders.py), in flight from frame 7-25 on
  raise_frame, arrays = np.full(E, -1, np.int32), None
  for e in range(E):
    choice.env = e
    conv = rendering.ObservationToArray(mapping, dtype=np.float32)
    g = make()
    obs, _, _ = g.its_showtime()
    for f in range(T + 1):
      if f:
        if g.game_over:
          g = make()
          obs, _, _ = g.its_showtime()
        else:
          obs, _, _ = g.play(int(binding.action_hash(SEED, e, f - 1) % n_actions))
      try:
        arr = conv(obs)
      except RuntimeError:
        raise_frame[e] = f
        break
      if arrays is None:
        arrays = np.zeros((T + 1, E) + arr.shape, np.float32)
      arrays[f, e] = arr
  np.savez_compressed(out_path('marauders_to_array'), template=np.frombuffer(b'marauders', np.uint8), seed=np.array([SEED], np.uint64),
                      mapping_chars=np.frombuffer(''.join(mapping).encode(), np.uint8), mapping_values=np.array(list(mapping.values()), np.float32),
                      arrays=arrays, raise_frame=raise_frame)
  print('marauders_to_array: %d of %d environments raised within %d steps' % (int((raise_frame >= 0).sum()), E, T), flush=True)


def fixed_crop_overhang():
  make, _ = ref_live._import_game('scrolly_maze_L0')
  from pycolab import cropping
  out = {}
  for tag, (corner, rows, cols, pad) in {'overhang': ((-1, 3), 5, 7, None), 'inside': ((2, 3), 5, 7, None), 'padded': ((-1, 3), 5, 7, ' ')}.items():
    g = make()
    cr = cropping.FixedCropper(corner, rows, cols, pad)
    cr.set_engine(g)
    obs, _, _ = g.its_showtime()
    try:
      cr.crop(obs)
      out[tag] = 0
    except RuntimeError:
      out[tag] = 1
  assert out == {'overhang': 1, 'inside': 0, 'padded': 0}, out
  np.savez_compressed(out_path('fixed_crop_overhang'), template=np.frombuffer(b'scrolly_maze_L0', np.uint8),
                      corner=np.array([-1, 3]), rows=np.array([5]), cols=np.array([7]), raises=np.array([1]))
  print('fixed_crop_overhang: the reference raises at frame 0', flush=True)


def main():
  only = set(a for a in sys.argv[1:] if not a.startswith('-'))  # fixture names to (re)generate; empty = all
  want = lambda name: not only or name in only
  for name in ('walkers_scroll_always', 'walkers_scroll_margins', 'walkers_scroll_groups', 'walkers_room', 'walkers_scroll_disagree'):
    if want(name):
      walkers(name, E=128, T=320)
  from oracle import custom_levels
  for name in custom_levels.WAREHOUSE_OPEN_NAMES:
    if want(name):
      warehouse_open(name, E=96, T=256)
  if want('marauders_to_array'):
    marauders_to_array(E=32, T=96)
  if want('fixed_crop_overhang'):
    fixed_crop_overhang()


if __name__ == '__main__':
  main()
