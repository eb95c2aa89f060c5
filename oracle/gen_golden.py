#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Golden traces from the *imported reference itself*.

Imports google-deepmind/pycolab from /root/reference (read-only), runs the
unchanged example games on seeded action tapes with the episode policy of
SURVEY.md section 8(d) (an environment whose episode ended is rebuilt with
make_game()+its_showtime() at the next step), and records every step's
board, reward (+ "is None" flag), discount, game_over and sprite states.
Occluded layers are asserted to equal `board == ord(c)` for every character
at generation time (rendering.py:177-179), so tests derive expected layers
from the recorded boards.

Run here (CPU container):  python oracle/gen_golden.py
Output: tests/golden/traces/*.npz  (committed; the GPU box has no reference).
"""
import os
import sys

if os.environ.get('PYTHONHASHSEED') != '0':
  # hello_world.py builds its update schedule from a set of characters: pin the
  # string hash seed so that templates and traces come out the same every run
  os.environ['PYTHONHASHSEED'] = '0'
  os.execv(sys.executable, [sys.executable] + sys.argv)
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('PCX_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
warnings.filterwarnings('ignore', category=DeprecationWarning)

NONE = -1


def tape(rng, policy, T, n_ordinary, quit_action):
  """One environment's action tape (int32 [T]; -1 = None)."""
  a = np.zeros(T, np.int32)
  cur = rng.randint(n_ordinary)
  for t in range(T):
    if policy == 0:
      cur = rng.randint(n_ordinary)
    elif policy == 1:
      if rng.rand() > 0.85:
        cur = rng.randint(n_ordinary)
    elif policy == 2:
      r = rng.rand()
      if r < 0.01:
        a[t] = quit_action
        continue
      if r < 0.03:
        a[t] = NONE
        continue
      if r < 0.05:
        a[t] = quit_action + 1 + rng.randint(3)  # garbage actions
        continue
      if r > 0.8:
        cur = rng.randint(n_ordinary)
    else:
      if rng.rand() > 0.6:
        cur = rng.randint(max(1, n_ordinary - 1))
    a[t] = cur
  return a


def seek_coin_action(game, rng):
  """Greedy BFS towards the nearest coin in world coordinates (test tapes
  that actually collect coins and scroll; uses reference internals)."""
  import collections
  walls = game.things['#'].whole_pattern
  coins = game.things['@'].whole_pattern
  corner = game.things['#']._northwest_corner
  vp = game.things['P'].virtual_position
  start = (vp[0] + corner[0], vp[1] + corner[1])
  if rng.rand() < 0.08 or not coins.any():
    return rng.randint(5)
  moves = [(-1, 0), (1, 0), (0, -1), (0, 1)]
  seen = {start: None}
  q = collections.deque([start])
  while q:
    cur = q.popleft()
    if coins[cur] and cur != start:
      while seen[cur][0] != start:
        cur = seen[cur][0]
      return seen[cur][1]
    for a, (dr, dc) in enumerate(moves):
      nxt = (cur[0] + dr, cur[1] + dc)
      if (0 <= nxt[0] < walls.shape[0] and 0 <= nxt[1] < walls.shape[1]
          and not walls[nxt] and nxt not in seen):
        seen[nxt] = (cur, a)
        q.append(nxt)
  return rng.randint(5)


def sprite_states(game, sprite_chars):
  out = np.zeros((len(sprite_chars), 5), np.int16)
  for i, c in enumerate(sprite_chars):
    s = game.things[c]
    vp = getattr(s, 'virtual_position', s.position)
    out[i] = (s.position[0], s.position[1], vp[0], vp[1], int(bool(s.visible)))
  return out


UNOCCLUDED = False  # set by run(..., unoccluded=True)


def record(obs, reward, discount, game, chars, sprite_chars):
  if not UNOCCLUDED:
    for c in chars:  # occluded layers are board == c
      assert np.array_equal(obs.layers[c], obs.board == ord(c)), c
  assert set(obs.layers.keys()) == set(chars)
  if isinstance(reward, float):  # a game whose entities add Python floats (directives_float_rewards; plot.py:200-226)
    assert float(np.float32(reward)) == reward, reward  # (the fixture's values are exact in the float32 lane)
  return (obs.board.copy(), 0 if reward is None else (reward if isinstance(reward, float) else int(reward)),
          0 if reward is None else 1, float(discount), int(game.game_over),
          sprite_states(game, sprite_chars),
          np.stack([obs.layers[c] for c in chars]).astype(np.uint8) if UNOCCLUDED else None)


def template_sprite_chars(template_name):
  sys.path.insert(0, ROOT)
  from pycolab_amd.compiler import GameTemplate
  t = GameTemplate.load(os.path.join(ROOT, 'tests', 'golden', 'templates', template_name + '.npz'))
  return [chr(s['ch']) for s in t.sprites]


class ChoicePatch(object):
  """Replaces numpy's global-RNG `np.random.choice` (used by
  extraterrestrial_marauders.py:253) with the counter-based draw that the
  oracle and the kernel implement: arr[hash(seed ^ SALT, env, draw#) % len]."""
  SALT = 0x4D415241554445

  def __init__(self, seed):
    from oracle import binding
    self._hash = binding.action_hash
    self.seed, self.env, self.draws = seed, 0, {}

  def __call__(self, arr):
    n = self.draws.get(self.env, 0)
    self.draws[self.env] = n + 1
    return arr[self._hash(self.seed ^ self.SALT, self.env, n) % len(arr)]


def make_reference_croppers(specs):
  from pycolab import cropping
  out = []
  for sp in specs:
    if sp['kind'] == 'fixed':
      out.append(cropping.FixedCropper(tuple(sp['top_left']), sp['rows'], sp['cols'], sp['pad_char']))
    else:
      out.append(cropping.ScrollingCropper(
          sp['rows'], sp['cols'], list(sp['to_track']), pad_char=sp['pad_char'],
          scroll_margins=tuple(sp['scroll_margins']),
          initial_offset=None if sp['initial_offset'] is None else tuple(sp['initial_offset']),
          saccade=sp['saccade']))
  return out


def S(rows, cols, to_track, pad_char=None, scroll_margins=(2, 3), initial_offset=None, saccade=True):
  return dict(kind='scrolling', rows=rows, cols=cols, to_track=list(to_track), pad_char=pad_char,
              scroll_margins=list(scroll_margins), initial_offset=initial_offset, saccade=saccade)


def F(top_left, rows, cols, pad_char=None):
  return dict(kind='fixed', top_left=list(top_left), rows=rows, cols=cols, pad_char=pad_char)


POST = {  # keyed by trace name: post-processor specs (JSON-able)
    'marauders': [
        dict(kind='repaint', mapping={'a': '^', 'b': '^', 'c': '^', 'd': '^', 'y': '|', 'z': '|'}),
        dict(kind='to_array', mapping={' ': [0, 0, 0], 'B': [400, 50, 30], 'P': [0, 999, 0], 'X': [999, 999, 999],
                                       'a': [0, 999, 999], 'b': [0, 999, 999], 'c': [0, 999, 999],
                                       'd': [0, 999, 999], 'y': [7, 8, 9], 'z': [7, 8, 9]},
             dtype='float32', permute=[1, 2, 0]),
        dict(kind='to_array', mapping={' ': 0, 'B': 1, 'P': 2, 'X': 3, 'a': 4, 'b': 4, 'c': 4, 'd': 4, 'y': 5, 'z': 5},
             dtype='uint8', permute=[1, 0]),
        dict(kind='features', layers='PXBq', permute=[2, 0, 1]),
    ],
    'warehouse_L1': [
        dict(kind='repaint', mapping={c: 'x' for c in '0123456789'}),
        dict(kind='features', layers='P_#12', permute=None),
        dict(kind='to_array', mapping={c: [i, 2 * i] for i, c in enumerate(' #._1234567PX')}, dtype='int32',
             permute=None),
    ],
}
# games only the table-driven kernel steps (round 4: its render loop writes the planar feature array itself): the
# reference's ObservationToFeatureArray on prefab walkers in two scrolling groups, z-order directives, an unshipped
# marauders board -- 'q' / '~' are characters the games do not have (planes of zeros)
POST['walkers_scroll_groups'] = [dict(kind='features', layers='P%Qa# bq', permute=None)]
POST['directives_z_order'] = [dict(kind='features', layers='cbaD.~', permute=None),
                              dict(kind='to_array', mapping={'.': 0.0, 'D': 0.5, 'a': 1.0, 'b': 2.0, 'c': 4.0}, dtype='float32', permute=None)]
POST['marauders_custom_A'] = [dict(kind='features', layers='XPB ayq', permute=None)]

POST_EVERY = 8  # frames between recorded post-processor outputs

# crop, THEN post-process (human_ui.py:252-265; better_scrolly_maze.py:237-247 into rendering.py:545-661): the
# post-processor of the second element applied to what cropper #first returned, recorded at the POST frames
CROP_POST = {  # keyed by trace name: [(index into CROPPERS[name], post-processor spec)]
    'better_scrolly_maze_L0': [(0, dict(kind='features', layers='P@#abc +', permute=None)),
                               (0, dict(kind='features', layers='@P#a', permute=[1, 2, 0])),
                               (1, dict(kind='features', layers='P#', permute=None))],
    'warehouse_L1': [(1, dict(kind='features', layers='P.#1_x', permute=None)),     # 7 x 7 window, padded with '.': 49 cells
                     (3, dict(kind='features', layers='XP _', permute=[1, 2, 0]))],  # 3 x 9, padded with ' '
    'marauders': [(1, dict(kind='features', layers='aPyB |', permute=None))],        # 9 x 11 = 99 cells, egocentric on a / y / P
}


def make_reference_post(spec):
  from pycolab import rendering
  if spec['kind'] == 'repaint':
    return rendering.ObservationCharacterRepainter(dict(spec['mapping']))
  if spec['kind'] == 'features':
    return rendering.ObservationToFeatureArray(list(spec['layers']), permute=spec['permute'])
  return rendering.ObservationToArray(dict(spec['mapping']), dtype=np.dtype(spec['dtype']), permute=spec['permute'])


CROPPERS = {  # keyed by trace name

    'scrolly_maze_L0': [S(5, 11, 'P', ' ', (1, 2)), S(7, 9, 'aP', None, (2, 3)), F((-2, -3), 8, 20, '#'),
                        S(5, 5, 'cb', '.', (None, None), (1, -1), False), S(3, 9, '@P', ' ', (1, 3))],
    'warehouse_L1': [S(5, 5, 'P', None, (1, 1)), S(7, 7, '3P', '.', (2, 2), (0, 1)), F((2, 3), 5, 6),
                     S(3, 9, 'XP', ' ', (1, 2)), F((8, 9), 6, 7, '#')],
    'warehouse_custom_C': [S(5, 7, 'P', None, (1, 2)), F((4, 8), 6, 9, '#'), S(3, 5, '2P', '.', (None, None))],
    'marauders': [S(7, 15, 'P', None, (2, 4)), S(9, 11, 'ayP', ' ', (None, None)), F((10, -5), 8, 20, 'B'),
                  S(5, 9, 'X', ' ', (1, 3), None, False)],
}


ONLY = set(a for a in sys.argv[1:] if not a.startswith('-'))  # trace names to (re)generate; empty = all


def run(name, make_game, E, T, n_ordinary, quit_action, seed, template_name, seeker=False, choice=None,
        unoccluded=False, ref_action=None, tapes=None, before_play=None):
  global UNOCCLUDED
  if ONLY and name not in ONLY:
    return
  UNOCCLUDED = unoccluded
  boards, rewards, rsets, discounts, dones, sprites, layers = [], [], [], [], [], [], []
  specs = CROPPERS.get(name, [])
  crops = [[] for _ in specs]
  post_specs = POST.get(name, [])
  posts = [[] for _ in post_specs]
  crop_post_specs = CROP_POST.get(name, [])
  crop_posts = [[] for _ in crop_post_specs]
  actions = np.zeros((T, E), np.int32)
  chars = sprite_chars = None
  for e in range(E):
    rng = np.random.RandomState(seed * 1000 + e)
    actions[:, e] = tapes(rng, T) if tapes else tape(rng, e % 4, T, n_ordinary, quit_action)
    if choice is not None:
      choice.env = e
    game = make_game()
    rec = []
    croppers = make_reference_croppers(specs)
    env_crops = [[] for _ in specs]

    last_crop = [None] * len(specs)

    def crop_all(obs):
      for i, cr in enumerate(croppers):
        out = cr.crop(obs)
        for c in chars:  # cropped layers stay board == c (pad included)
          assert np.array_equal(out.layers[c], out.board == ord(c)), (name, i, c)
        env_crops[i].append(out.board.copy())
        last_crop[i] = out

    crop_post_objs = [make_reference_post(sp) for _, sp in crop_post_specs]
    env_crop_posts = [[] for _ in crop_post_specs]

    def crop_post_all():
      for i, ((ci, sp), po) in enumerate(zip(crop_post_specs, crop_post_objs)):
        env_crop_posts[i].append(np.array(po(last_crop[ci])).copy())

    post_objs = [make_reference_post(sp) for sp in post_specs]
    env_posts = [[] for _ in post_specs]

    def post_all(obs):
      for i, (sp, po) in enumerate(zip(post_specs, post_objs)):
        out = po(obs)
        if sp['kind'] == 'repaint':
          keys = sorted(out.layers)
          for c in keys:
            assert np.array_equal(out.layers[c], out.board == ord(c))
          env_posts[i].append(out.board.copy())
        else:
          env_posts[i].append(np.array(out).copy())

    for cr in croppers:
      cr.set_engine(game)
    obs, r, d = game.its_showtime()
    if chars is None:
      chars = sorted(obs.layers.keys())
      # template order of sprites = engine insertion (update schedule) order
      sprite_chars = template_sprite_chars(template_name)
    rec.append(record(obs, r, d, game, chars, sprite_chars))
    crop_all(obs)
    post_all(obs)
    crop_post_all()
    for t in range(T):
      if game.game_over:
        game = make_game()
        for cr in croppers:
          cr.set_engine(game)  # a new Engine: scrolling croppers start over (cropping.py:378-391)
        obs, r, d = game.its_showtime()
      else:
        if seeker and e % 8 >= 5:
          actions[t, e] = seek_coin_action(game, rng)
        a = int(actions[t, e])
        if before_play is not None and a != NONE:
          before_play(game, a)
        obs, r, d = game.play(None if a == NONE else (ref_action(a) if ref_action else a))
      rec.append(record(obs, r, d, game, chars, sprite_chars))
      crop_all(obs)
      if (t + 1) % POST_EVERY == 0:
        post_all(obs)
        crop_post_all()
    for i in range(len(post_specs)):
      posts[i].append(env_posts[i])
    for i in range(len(crop_post_specs)):
      crop_posts[i].append(env_crop_posts[i])
    for i in range(len(specs)):
      crops[i].append(env_crops[i])
    boards.append([x[0] for x in rec]); rewards.append([x[1] for x in rec])
    rsets.append([x[2] for x in rec]); discounts.append([x[3] for x in rec])
    dones.append([x[4] for x in rec]); sprites.append([x[5] for x in rec])
    if unoccluded:
      layers.append([x[6] for x in rec])
  sw = lambda x, dt: np.ascontiguousarray(np.swapaxes(np.array(x, dtype=dt), 0, 1))
  out_root = os.environ.get('PCX_GOLDEN_OUT') or os.path.join(ROOT, 'tests', 'golden')  # PCX_GOLDEN_OUT: regenerate elsewhere (tests/test_fixture_reproducibility.py)
  os.makedirs(os.path.join(out_root, 'traces'), exist_ok=True)
  path = os.path.join(out_root, 'traces', name + '.npz')
  import json
  extra = {'crop_%d' % i: sw(crops[i], np.uint8) for i in range(len(specs))}
  extra['crop_specs'] = np.frombuffer(json.dumps(specs).encode(), np.uint8)
  for i in range(len(post_specs)):
    extra['post_%d' % i] = np.ascontiguousarray(np.swapaxes(np.array(posts[i]), 0, 1))  # [frames, E, ...]
  extra['post_specs'] = np.frombuffer(json.dumps(post_specs).encode(), np.uint8)
  for i in range(len(crop_post_specs)):
    extra['crop_post_%d' % i] = np.ascontiguousarray(np.swapaxes(np.array(crop_posts[i]), 0, 1))  # [frames, E, ...]
  if crop_post_specs:
    extra['crop_post_specs'] = np.frombuffer(json.dumps([[ci, sp] for ci, sp in crop_post_specs]).encode(), np.uint8)
  extra['post_every'] = np.array([POST_EVERY])
  if unoccluded:
    extra['layers'] = sw(layers, np.uint8)  # [T+1, E, L, R, C]
  np.savez_compressed(
      path, template=np.frombuffer(template_name.encode(), np.uint8),
      chars=np.array([ord(c) for c in chars], np.uint8),
      sprite_chars=np.array([ord(c) for c in sprite_chars], np.uint8),
      actions=actions, boards=sw(boards, np.uint8),
      reward=sw(rewards, np.float32 if any(isinstance(r, float) for env in rewards for r in env) else np.int32),
      reward_set=sw(rsets, np.uint8), discount=sw(discounts, np.float32),
      done=sw(dones, np.uint8), sprites=sw(sprites, np.int16), **extra)
  nd = int(np.array(dones).sum())
  nr = int(np.array(rsets).sum())
  print('wrote %s: E=%d T=%d episodes_ended=%d rewards=%d size=%d' % (
      path, E, T, nd, nr, os.path.getsize(path)))


def main():
  import numpy.random
  from pycolab.examples import scrolly_maze, warehouse_manager, hello_world, extraterrestrial_marauders
  for level in (0, 1, 2):
    run('scrolly_maze_L%d' % level, lambda: scrolly_maze.make_game(level),
        E=32, T=192, n_ordinary=5, quit_action=5, seed=7 + level,
        template_name='scrolly_maze_L%d' % level, seeker=True)
  sys.path.insert(0, ROOT)
  from oracle import custom_levels
  from pycolab import ascii_art as ref_ascii_art
  from pycolab.prefab_parts import drapes as ref_drapes
  for i, name in enumerate(custom_levels.NAMES):  # other board shapes / sprite sets / z-orders
    run(name, lambda: custom_levels.make_game(name, scrolly_maze, ref_ascii_art, ref_drapes),
        E=24, T=160, n_ordinary=5, quit_action=5, seed=81 + i, template_name=name, seeker=True)
  for i, name in enumerate(custom_levels.WAREHOUSE_NAMES):
    run(name, lambda: custom_levels.make_warehouse(name, warehouse_manager, ref_ascii_art),
        E=24, T=160, n_ordinary=5, quit_action=5, seed=71 + i, template_name=name)
  for level in (0, 1, 2):
    run('warehouse_L%d' % level, lambda: warehouse_manager.make_game(level),
        E=32, T=192, n_ordinary=5, quit_action=5, seed=17 + level,
        template_name='warehouse_L%d' % level)
  from pycolab.examples import better_scrolly_maze as bsm
  for level in (0, 1, 2):
    name = 'better_scrolly_maze_L%d' % level
    # the example's own make_croppers(level) (better_scrolly_maze.py:224-247), as specs
    CROPPERS[name] = [S(10, 30, 'P', None, (2, 3), list(bsm.STARTER_OFFSET[level])),
                      S(7, 10, 'c', ' ', (None, 3)),
                      F(bsm.TEASER_CORNER[level], 12, 20, ' ')]
    run(name, lambda: bsm.make_game(level), E=12, T=160, n_ordinary=5, quit_action=5, seed=61 + level,
        template_name=name)
  for i, name in enumerate(custom_levels.BETTER_NAMES):  # boards without a compiled kernel instance, with croppers
    art = custom_levels.better_art(name)
    rows, cols = len(art), len(art[0])
    CROPPERS[name] = [S(7, 11, 'P', None, (2, 3)), S(5, 9, 'bP', ' ', (None, None), (1, -2)),
                      F((rows - 4, cols - 6), 6, 10, '#')]
    run(name, lambda: custom_levels.make_better_scrolly(name, bsm, ref_ascii_art), E=12, T=160, n_ordinary=5,
        quit_action=5, seed=161 + i, template_name=name)
  run('hello_world', hello_world.make_game, E=16, T=96, n_ordinary=4, quit_action=4, seed=27,
      template_name='hello_world')
  for i, name in enumerate(custom_levels.HELLO_NAMES):
    run(name, lambda: custom_levels.make_hello(name, hello_world, ref_ascii_art), E=16, T=96, n_ordinary=4,
        quit_action=4, seed=141 + i, template_name=name)
  # prefab-only scenarios built from the reference's own test entities
  sys.path.insert(0, ROOT)
  from oracle import walker_scenarios
  from pycolab import ascii_art as ref_art
  from pycolab.tests import test_things as tt
  names = walker_scenarios.MOTION_NAMES
  seeds = {'walkers_room': 51, 'walkers_scroll_always': 52, 'walkers_scroll_margins': 53, 'walkers_scroll_groups': 54, 'walkers_hidden': 55}
  for name, spec in sorted(walker_scenarios.SCENARIOS.items()):
    i = seeds[name] - 51  # fixed per scenario: adding a scenario must not move the others' tapes
    if spec['kind'] == 'scroll2':  # two scrolling groups, one action field each
      fields = {}
      for w in spec['worlds']:
        for ch in list(w['scrollies']) + list(w['walkers']):
          fields[ch] = w['field']
      ref_action = lambda a, fields=fields: {ch: names[min((a >> sh) & mk, 8)] for ch, (sh, mk) in fields.items()}
      tapes = lambda rng, T, n=spec['n_fields']: sum(
          (walker_scenarios.field_tape(rng, T, True) << (4 * f) for f in range(n)), np.zeros(T, np.int32)).astype(np.int32)
      run(name, lambda spec=spec: walker_scenarios.build(spec, ref_art, tt.TestMazeWalker, tt.TestScrolly, False),
          E=24, T=160, n_ordinary=9, quit_action=99, seed=51 + i, template_name=name,
          ref_action=ref_action, tapes=tapes)
      continue
    chars = sorted(spec['walkers'])
    if spec['n_fields']:
      fields = {ch: spec['walkers'][ch]['field'] for ch in chars}
      ref_action = lambda a, fields=fields: {ch: names[min((a >> sh) & mk, 8)] for ch, (sh, mk) in fields.items()}
      tapes = lambda rng, T, n=spec['n_fields']: sum(
          (walker_scenarios.field_tape(rng, T) << (4 * f) for f in range(n)), np.zeros(T, np.int32)).astype(np.int32)
    else:
      ref_action = lambda a: names[min(a, 8)]
      tapes = lambda rng, T, c=spec.get('cardinal_only', False): walker_scenarios.field_tape(rng, T, c)
    run(name, lambda spec=spec: walker_scenarios.build(spec, ref_art, tt.TestMazeWalker, tt.TestScrolly, False),
        E=24, T=160, n_ordinary=9, quit_action=99, seed=51 + i, template_name=name,
        ref_action=ref_action, tapes=tapes)

  # Plot directives (add_reward / terminate_episode(discount) / change_z_order)
  # issued by the reference's own test entities: tests/engine_test.py:169-295
  from oracle import directive_scenarios as ds
  seeds = {'directives_reward_discount': 151, 'directives_z_order': 152, 'directives_two_discounts': 153, 'directives_float_rewards': 154}  # fixed per scenario
  for name, spec in sorted(ds.SCENARIOS.items()):
    run(name, lambda spec=spec: ds.build_reference(spec, ref_art, tt), E=24, T=160, n_ordinary=9, quit_action=99,
        seed=seeds[name], template_name=name, ref_action=lambda a, spec=spec: ds.reference_action(spec, a),
        tapes=lambda rng, T, spec=spec: ds.tape(spec, rng, T),
        before_play=lambda game, a, spec=spec: ds.inject(spec, game, a, tt))

  # occlusion_in_layers=False variants: the example files do not expose the
  # flag, so the call they make into ascii_art is wrapped (files unchanged).
  import functools
  from pycolab import ascii_art as ref_ascii_art
  real = ref_ascii_art.ascii_art_to_game
  ref_ascii_art.ascii_art_to_game = functools.partial(real, occlusion_in_layers=False)
  try:
    run('scrolly_maze_L1_unoccluded', lambda: scrolly_maze.make_game(1), E=16, T=128, n_ordinary=5,
        quit_action=5, seed=41, template_name='scrolly_maze_L1_unoccluded', seeker=True, unoccluded=True)
    run('warehouse_L0_unoccluded', lambda: warehouse_manager.make_game(0), E=16, T=128, n_ordinary=5,
        quit_action=5, seed=42, template_name='warehouse_L0_unoccluded', unoccluded=True)
    for i, base in enumerate(custom_levels.UNOCCLUDED):
      run(base + '_unoccluded', lambda: custom_levels.make_game(base, scrolly_maze, ref_ascii_art, ref_drapes),
          E=12, T=128, n_ordinary=5, quit_action=5, seed=91 + i, template_name=base + '_unoccluded', seeker=True,
          unoccluded=True)
    patch_u = ChoicePatch(seed=0x5EED)
    numpy.random.choice = patch_u
    run('marauders_unoccluded', extraterrestrial_marauders.make_game, E=16, T=192, n_ordinary=4,
        quit_action=4, seed=43, template_name='marauders_unoccluded', choice=patch_u, unoccluded=True)
  finally:
    ref_ascii_art.ascii_art_to_game = real
  patch = ChoicePatch(seed=0x5EED)   # the engines under test get the same seed (template param)
  numpy.random.choice = patch
  run('marauders', extraterrestrial_marauders.make_game, E=32, T=256, n_ordinary=4, quit_action=4,
      seed=37, template_name='marauders', choice=patch)
  for i, name in enumerate(custom_levels.MARAUDERS_NAMES):
    patch_c = ChoicePatch(seed=0x5EED)
    numpy.random.choice = patch_c
    run(name, lambda: custom_levels.make_marauders(name, extraterrestrial_marauders, ref_ascii_art),
        E=24, T=224, n_ordinary=4, quit_action=4, seed=131 + i, template_name=name, choice=patch_c)


if __name__ == '__main__':
  main()
