"""TEST INFRASTRUCTURE: games whose entities issue Plot directives --
add_reward, terminate_episode(discount), change_z_order -- the calls the
reference's tests/engine_test.py:169-295 injects into its test entities with
`tt.pre_update`.  The same description builds

  * the reference game (its own `tt.TestMazeWalker` / `tt.TestSprite` /
    `tt.TestDrape`; `inject()` registers, before every `play()`, the
    `tt.pre_update` callables the action's directive fields select), and
  * the pycolab_amd twin (tabled prefabs carrying the calls as data).

An action packs, per entity, a 4-bit motion field (walkers) and a 2-bit
directive field.
"""

# name -> dict(art, beneath, z_order, schedule, entities={ch: dict(kind, motion=(shift, mask)|None,
#   directive=(shift, mask), calls={value: [call, ...]}, **ctor kwargs)})
SCENARIOS = {
    # engine_test.py:244-295 testChangingZOrdering, on a board with room to
    # walk: three walkers that pile up and a drape, all re-ordered at run time
    'directives_z_order': dict(
        art=['.......',
             '.abc.D.',
             '...D...'],
        beneath='.', z_order='aDbc', schedule=[['a', 'b'], ['c', 'D']],
        entities={
            'a': dict(kind='walker', impassable='', confined=True, motion=(0, 15), directive=(12, 3),
                      calls={1: [('change_z_order', 'a', 'c')], 2: [('change_z_order', 'a', None)],
                             3: [('change_z_order', 'a', 'D'), ('add_reward', 2)]}),
            'b': dict(kind='walker', impassable='', confined=True, motion=(4, 15), directive=(14, 3),
                      calls={1: [('change_z_order', 'b', 'c')], 2: [('change_z_order', 'b', None)],
                             3: [('change_z_order', 'D', 'b'), ('change_z_order', 'b', 'a')]}),
            'c': dict(kind='walker', impassable='', confined=True, motion=(8, 15), directive=(16, 3),
                      calls={1: [('change_z_order', 'c', None)], 2: [('change_z_order', 'c', 'a')],
                             3: [('add_reward', -1)]}),
            'D': dict(kind='drape', motion=None, directive=(18, 3),
                      calls={1: [('change_z_order', 'D', None)], 2: [('change_z_order', 'D', 'c')],
                             3: [('terminate_episode', 0.25), ('add_reward', 10)]}),
        }),
    # engine_test.py:169-242 testRewardAndEpisodeEndWith{Default,Custom}Discount:
    # two sprites that only talk to the Plot (integer rewards instead of strings)
    'directives_reward_discount': dict(
        art=['.........',
             '...Q.R...',
             '.........'],
        beneath='.', z_order='QR', schedule=[['Q', 'R']],
        entities={
            'Q': dict(kind='sprite', motion=None, directive=(0, 3),
                      calls={1: [('add_reward', 5)], 2: [('terminate_episode',)], 3: [('terminate_episode', 0.5)]}),
            'R': dict(kind='sprite', motion=None, directive=(2, 3),
                      calls={1: [('add_reward', 7)], 2: [('add_reward', 11)], 3: [('add_reward', -3), ('add_reward', 4)]}),
        }),
    # plot.py:176-198: every terminate_episode(discount) of a step overwrites the discount -- the LAST one stands,
    # within one entity's update and across entities (Q updates before R).  No other fixture ends an episode twice in
    # one step (oracle/mutants.py: `later_terminate_keeps_the_first_discount` survived them all).
    'directives_two_discounts': dict(
        art=['.......',
             '..Q.R..',
             '.......'],
        beneath='.', z_order='QR', schedule=[['Q', 'R']],
        entities={
            'Q': dict(kind='sprite', motion=None, directive=(0, 3),
                      calls={1: [('terminate_episode', 0.5)], 2: [('terminate_episode', 0.75), ('terminate_episode', 0.25)],
                             3: [('add_reward', 3), ('terminate_episode', 0.125)]}),
            'R': dict(kind='sprite', motion=None, directive=(2, 3),
                      calls={1: [('terminate_episode', 0.0625)], 2: [('add_reward', 1)], 3: [('terminate_episode',)]}),
        }),
    # plot.py:200-226: add_reward sums anything `+=`-able -- here Python floats, as examples/ordeal.py:123, 187-190 adds them
    # (round 6: pcx_template::reward_is_float, a float32 reward lane).  The values are dyadic, so the reference's double
    # sums and the lane's float32 sums agree to the bit; a walker so that the board moves too.
    'directives_float_rewards': dict(
        art=['.........',
             '..Q.w.R..',
             '.........'],
        beneath='.', z_order='QwR', schedule=[['Q', 'w'], ['R']], float_rewards=True,
        entities={
            'Q': dict(kind='sprite', motion=None, directive=(0, 3),
                      calls={1: [('add_reward', 0.5)], 2: [('add_reward', -1.25), ('add_reward', 2.0)], 3: [('terminate_episode', 0.5), ('add_reward', 0.125)]}),
            'w': dict(kind='walker', impassable='', confined=True, motion=(4, 15), directive=(8, 3),
                      calls={1: [('add_reward', 2.75)], 2: [('add_reward', -0.375)], 3: [('add_reward', 1)]}),
            'R': dict(kind='sprite', motion=None, directive=(2, 3),
                      calls={1: [('add_reward', 1024.5)], 2: [('add_reward', -0.0625)], 3: [('terminate_episode',)]}),
        }),
}

MOTION_NAMES = ['n', 'ne', 'e', 'se', 's', 'sw', 'w', 'nw', 'stay']


def build_twin(spec, ascii_art, tabled):
  """The game with pycolab_amd's tabled prefabs."""
  P = ascii_art.Partial
  sprites, drapes = {}, {}
  for ch, e in spec['entities'].items():
    kw = dict(directive_field=e['directive'], directives=e['calls'])
    if spec.get('float_rewards'):
      kw['float_rewards'] = True
    if e['kind'] == 'walker':
      sprites[ch] = P(tabled.TabledWalker, impassable=e['impassable'], confined_to_board=e['confined'],
                      action_field=e['motion'], **kw)
    elif e['kind'] == 'sprite':
      sprites[ch] = P(tabled.StaticSprite, **kw)
    else:
      drapes[ch] = P(tabled.StaticDrape, **kw)
  return ascii_art.ascii_art_to_game(spec['art'], spec['beneath'], sprites=sprites, drapes=drapes,
                                     update_schedule=spec['schedule'], z_order=spec['z_order'])


def build_reference(spec, ascii_art, tt):
  """The game with the reference's own test entities."""
  P = ascii_art.Partial
  sprites, drapes = {}, {}
  for ch, e in spec['entities'].items():
    if e['kind'] == 'walker':
      sprites[ch] = P(tt.TestMazeWalker, impassable=e['impassable'], confined_to_board=e['confined'])
    elif e['kind'] == 'sprite':
      sprites[ch] = tt.TestSprite
    else:
      drapes[ch] = tt.TestDrape
  return ascii_art.ascii_art_to_game(spec['art'], spec['beneath'], sprites=sprites, drapes=drapes,
                                     update_schedule=spec['schedule'], z_order=spec['z_order'])


def reference_action(spec, a):
  """The packed action as the dict of compass directions TestMazeWalker obeys."""
  out = {}
  for ch, e in spec['entities'].items():
    if e['motion'] is not None:
      out[ch] = MOTION_NAMES[min((a >> e['motion'][0]) & e['motion'][1], 8)]
  return out


def inject(spec, game, a, tt):
  """Registers the pre-update callables action `a` selects (tests/test_things.py:57-76)."""
  for ch, e in spec['entities'].items():
    sel = (a >> e['directive'][0]) & e['directive'][1]
    calls = e['calls'].get(sel)
    if not calls:
      continue

    def thing_to_do(actions, board, layers, backdrop, things, the_plot, calls=calls):
      for call in calls:
        if call[0] == 'next_chapter':
          the_plot.next_chapter = call[1]  # plot.py:299-324 is a property setter
        else:
          getattr(the_plot, call[0])(*call[1:])
    tt.pre_update(game, ch, thing_to_do)


def tape(spec, rng, T):
  """One environment's packed actions: persistent headings for the walkers,
  directive fields mostly 0."""
  import numpy as np
  out = np.zeros(T, np.int32)
  heading = {ch: rng.randint(9) for ch, e in spec['entities'].items() if e['motion'] is not None}
  for t in range(T):
    a = 0
    for ch, e in spec['entities'].items():
      if e['motion'] is not None:
        if rng.rand() < 0.35:
          heading[ch] = rng.randint(9)
        a |= heading[ch] << e['motion'][0]
      if rng.rand() < 0.3:
        a |= rng.randint(1, e['directive'][1] + 1) << e['directive'][0]
    out[t] = a
  return out


# Three compatible chapters (same 3x9 board; every character used one way) for
# storytelling.Story (oracle/gen_story_golden.py): each ends through a
# terminate_episode directive, some with a custom discount.
STORY = [
    dict(art=['.........', '...Q.R...', '.........'], beneath='.', z_order='QR', schedule=[['Q', 'R']],
         entities={
             'Q': dict(kind='sprite', motion=None, directive=(0, 3),
                       calls={1: [('add_reward', 5)], 2: [('terminate_episode',)], 3: [('terminate_episode', 0.5), ('add_reward', 1)]}),
             'R': dict(kind='sprite', motion=None, directive=(2, 3),
                       calls={1: [('add_reward', 7)], 2: [('add_reward', 11)], 3: [('add_reward', -3)]})}),
    dict(art=['.........', '.a.....b.', '....D....'], beneath='.', z_order='aDb', schedule=[['a', 'b'], ['D']],
         entities={
             'a': dict(kind='walker', impassable='', confined=True, motion=(4, 15), directive=(0, 3),
                       calls={1: [('change_z_order', 'a', 'b')], 3: [('add_reward', 2)]}),
             'b': dict(kind='walker', impassable='a', confined=True, motion=(8, 15), directive=(2, 3),
                       calls={2: [('change_z_order', 'b', None)]}),
             'D': dict(kind='drape', motion=None, directive=(12, 3),
                       calls={1: [('terminate_episode', 0.25), ('add_reward', 10)], 2: [('add_reward', 1)]})}),
    dict(art=['....Q....', '.........', '..c......'], beneath='.', z_order='cQ', schedule=[['c', 'Q']],
         entities={
             'c': dict(kind='walker', impassable='Q', confined=False, motion=(4, 15), directive=(2, 3),
                       calls={3: [('add_reward', 100)]}),
             'Q': dict(kind='sprite', motion=None, directive=(0, 3),
                       calls={1: [('add_reward', 5)], 2: [('terminate_episode',)], 3: [('terminate_episode', 0.5), ('add_reward', 1)]})}),
]


# The same three chapters with entities that decide the story's order themselves
# (`the_plot.next_chapter = ...` from inside update(), as examples/ordeal.py:177-235 does;
# plot.py:299-324): jumps ahead, jumps back, "the story ends after this game".
import copy as _copy
STORY_JUMPS = _copy.deepcopy(STORY)
STORY_JUMPS[0]['entities']['Q']['calls'] = {1: [('add_reward', 5)], 2: [('next_chapter', 2), ('terminate_episode',)],
                                            3: [('terminate_episode', 0.5), ('add_reward', 1)]}
STORY_JUMPS[0]['entities']['R']['calls'] = {1: [('next_chapter', None)], 2: [('add_reward', 11)], 3: [('next_chapter', 1), ('add_reward', -3)]}
STORY_JUMPS[1]['entities']['D']['calls'] = {1: [('next_chapter', 0), ('terminate_episode', 0.25), ('add_reward', 10)], 2: [('add_reward', 1)]}
STORY_JUMPS[1]['entities']['b']['calls'] = {2: [('change_z_order', 'b', None)], 1: [('next_chapter', 2)]}
STORY_JUMPS[2]['entities']['Q']['calls'] = {1: [('next_chapter', 1), ('add_reward', 5)], 2: [('terminate_episode',)],
                                            3: [('next_chapter', 0), ('terminate_episode', 0.5), ('add_reward', 1)]}


def story_tape(rng, T):
  """Packed actions for a story run: every field of every chapter gets random bits."""
  import numpy as np
  out = np.zeros(T, np.int32)
  for t in range(T):
    a = rng.randint(9) << 4 | rng.randint(9) << 8
    if rng.rand() < 0.45:
      a |= rng.randint(1, 4)
    if rng.rand() < 0.4:
      a |= rng.randint(1, 4) << 2
    if rng.rand() < 0.25:
      a |= rng.randint(1, 3) << 12
    out[t] = a
  return out
