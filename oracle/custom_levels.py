"""TEST INFRASTRUCTURE: scrolly_maze levels that the reference does not ship.

The three shipped levels all have a 10x30 board, four sprites and the z-order
'abc@#P', so they exercise one shape of the step kernel only.  These levels use
the example file's own, unchanged classes (PlayerSprite, PatrollerSprite,
MazeDrape, CashDrape; scrolly_maze.py:238-357) with other board shapes, sprite
sets and z-orders.  `make_game(level, example, ascii_art, prefab_drapes)` builds
one the way scrolly_maze.make_game does (scrolly_maze.py:213-235), with the
example module and the pycolab packages passed in: oracle/gen_golden.py passes
the reference's, oracle/gen_templates.py passes pycolab_amd's.

The art is drawn once from a fixed seed (no randomness at test time).
"""
import numpy as np


def _draw(seed, rows, cols, board, sprites, wall_p, coin_p, mark_beneath):
  """A walled maze of rows x cols with random interior walls and coins, a
  board-corner mark such that `board` fits, and the sprites on free cells (the
  player inside the board window)."""
  rng = np.random.RandomState(seed)
  br, bc = board
  art = np.full((rows, cols), ' ', dtype='<U1')
  art[0, :] = art[-1, :] = art[:, 0] = art[:, -1] = '#'
  inner = rng.rand(rows - 2, cols - 2)
  art[1:-1, 1:-1][inner < wall_p] = '#'
  art[1:-1, 1:-1][(inner >= wall_p) & (inner < wall_p + coin_p)] = '@'
  cr = int(rng.randint(0, rows - br + 1))
  cc = int(rng.randint(0, cols - bc + 1))
  free = lambda r, c: art[r, c] == ' ' and (r, c) != (cr, cc)
  for ch in sprites:
    for _ in range(10000):
      if ch == 'P':  # the egocentric player starts inside the window, away from its rim
        r = int(rng.randint(cr + 1, cr + br - 1)); c = int(rng.randint(cc + 1, cc + bc - 1))
      else:
        r = int(rng.randint(1, rows - 1)); c = int(rng.randint(1, cols - 1))
      if 0 < r < rows - 1 and 0 < c < cols - 1 and free(r, c):
        art[r, c] = ch
        break
    else:
      raise RuntimeError('no room for sprite ' + ch)
  beneath = art[cr, cc] if art[cr, cc] in mark_beneath else mark_beneath[0]
  art[cr, cc] = '+'
  stars = np.full((br, bc), ' ', dtype='<U1')
  stars[rng.rand(br, bc) < 0.1] = '.'
  return [''.join(row) for row in art], [''.join(row) for row in stars], str(beneath)


# name -> (seed, maze rows, maze cols, board (rows, cols), sprites in update order, z_order)
# Board shapes: 18 dwords per board (less than one wavefront's 64), exactly 64,
# 60, and 10 (and 40, 36); one to six sprites; the player at the back, the
# middle and the front of the z-order.
SPECS = {
    'scrolly_custom_A': (101, 14, 25, (6, 12), 'aP', 'a@#P'),
    'scrolly_custom_B': (102, 20, 40, (8, 32), 'abcP', 'P#@cba'),
    'scrolly_custom_C': (103, 12, 20, (12, 20), 'P', '#P@'),
    'scrolly_custom_D': (104, 9, 31, (5, 8), 'bP', '@bP#'),
    'scrolly_custom_E': (105, 30, 24, (10, 24), 'cbP', 'c#Pb@'),
    'scrolly_custom_F': (106, 16, 40, (8, 20), 'abcdP', 'dcba@#P'),
    'scrolly_custom_G': (107, 18, 30, (9, 16), 'abcdeP', 'Pe@d#cba'),
    # round 6: the EXAMPLE'S OWN shape -- its 10x30 board, 'abcP' cast and z-order (scrolly_maze.py:212-242) -- around a maze
    # the reference does not ship: what a new entry of MAZES_ART looks like to the library.  Such a level takes the
    # instances pcx_scrolly_maze_step builds for it at run time (persistent workers and the cooperative shape with the
    # level's constants compiled in, csrc/pcx_scrolly_maze.hip jit), where the shipped levels take the ones in libpcx.so.
    'scrolly_custom_H': (108, 26, 64, (10, 30), 'abcP', 'abc@#P'),
}
NAMES = sorted(SPECS)
# also recorded with Engine(occlusion_in_layers=False): one, two and three sprites
UNOCCLUDED = ['scrolly_custom_A', 'scrolly_custom_C', 'scrolly_custom_E']


def level_art(name):
  seed, rows, cols, board, sprites, _ = SPECS[name]
  return _draw(seed, rows, cols, board, sprites, wall_p=0.22, coin_p=0.08, mark_beneath='# ')


def make_game(name, example, ascii_art, prefab_drapes):
  """scrolly_maze.make_game (scrolly_maze.py:213-235) for a level of SPECS."""
  _, _, _, _, sprites, z_order = SPECS[name]
  maze, stars, beneath = level_art(name)
  info = prefab_drapes.Scrolly.PatternInfo(maze, stars, board_northwest_corner_mark='+',
                                           what_lies_beneath=beneath)
  parts = {}
  for ch in sprites:
    cls = example.PlayerSprite if ch == 'P' else example.PatrollerSprite
    parts[ch] = ascii_art.Partial(cls, info.virtual_position(ch))
  return ascii_art.ascii_art_to_game(
      stars, what_lies_beneath=' ', sprites=parts,
      drapes={'#': ascii_art.Partial(example.MazeDrape, **info.kwargs('#')),
              '@': ascii_art.Partial(example.CashDrape, **info.kwargs('@'))},
      update_schedule=[['#'], list(sprites), ['@']], z_order=z_order)


# ---- warehouse_manager levels that the reference does not ship -------------------
# The example's own classes (BoxSprite, JudgeDrape, PlayerSprite;
# warehouse_manager.py:181-295) built the way warehouse_manager.make_game does
# (:139-178): two boxes on a 7x9 board, and all ten box characters on a 12x18
# board (twelve things, sixteen characters).
WAREHOUSE_ART = {
    'warehouse_custom_A': ['.........',
                           '.#######.',
                           '.# _   #.',
                           '.# 1 2 #.',
                           '.#  P _#.',
                           '.#######.',
                           '.........'],
    'warehouse_custom_B': ['..................',
                           '.################.',
                           '.#   _  #  _    #.',
                           '.# 1   2#   3 _ #.',
                           '.#   #     #    #.',
                           '.# _ # 4 5 #  6 #.',
                           '.#   #     #    #.',
                           '.#  7   P    8  #.',
                           '.# _    ##   _  #.',
                           '.#  9  _  0  _  #.',
                           '.################.',
                           '..................'],
    # shapes without a compiled instance of pcx_warehouse_step: its run-time-shape instances
    'warehouse_custom_C': ['...............',
                           '.#############.',
                           '.#  _   #  _ #.',
                           '.# 1  2    # #.',
                           '.#   ##  3   #.',
                           '.# _    P  4 #.',
                           '.#    #   _  #.',
                           '.#############.',
                           '...............'],
    'warehouse_custom_D': ['........',
                           '.######.',
                           '.# 1 _#.',
                           '.#  P #.',
                           '.######.',
                           '........'],
}
WAREHOUSE_NAMES = sorted(WAREHOUSE_ART)
# Warehouses WITHOUT walls around them, as RAISE fixtures only (oracle/gen_raise_golden.py; no golden trace: sooner or
# later a box reaches the last row or column and `layers['P'][row + 1, col]` is an IndexError, warehouse_manager.py:219-226).
# What they pin before that: numpy's NEGATIVE indices.  A box in row 0 asked to go south looks for the player at
# `layers['P'][-1, col]` -- the LAST row -- and in open_A the player starts exactly there: the box moves, pushed from the
# other side of the board (open_B: the same across the columns).  Boxes and player also walk off the open board
# (MazeWalkers are unconfined by default): a box out there has position (0, 0) for the Judge (warehouse_manager.py:248-250),
# which is a goal cell here.  (oracle/mutants.py: `negative_indices_do_not_wrap` survived every other fixture.)
WAREHOUSE_OPEN_ART = {
    'warehouse_open_A': ['_  1   _',
                         '     2  ',
                         ' _      ',
                         '3    _  ',
                         '    4   ',
                         '_  P   _'],
    'warehouse_open_B': ['_   _  ',
                         '  3    ',
                         '1     P',
                         '   _ 2 ',
                         ' 4     ',
                         '_     _'],
    # the same two situations with all four of the shipped levels' backdrop characters on the board (' ', '#', '.', '_'):
    # these are stepped by pcx_warehouse_step's run-time-shape instance (pcx_warehouse.hip: "the usual four backdrop-only
    # characters"), whose own copy of the index rules the two boards above -- three backdrop characters, stepped by
    # pcx_generic_step -- never reach.  Round 6 (VERDICT r5 missing #3b).
    'warehouse_open_C': ['_  1  . _',
                         '     2   ',
                         ' _  #    ',
                         '3     _  ',
                         ' .   4   ',
                         '_  P    _'],
    'warehouse_open_D': ['_   _  .',
                         '  3     ',
                         '1      P',
                         ' #  _ 2 ',
                         ' 4      ',
                         '_  .   _'],
}
WAREHOUSE_OPEN_NAMES = sorted(WAREHOUSE_OPEN_ART)
for _name, _art in list(WAREHOUSE_ART.items()) + list(WAREHOUSE_OPEN_ART.items()):
  assert len(set(len(_row) for _row in _art)) == 1, _name


def make_warehouse(name, example, ascii_art):
  """warehouse_manager.make_game (warehouse_manager.py:139-178) for WAREHOUSE_ART[name]."""
  art = WAREHOUSE_ART.get(name) or WAREHOUSE_OPEN_ART[name]
  boxes = [c for c in '1234567890' if c in ''.join(art)]
  sprites = {c: example.BoxSprite for c in boxes}
  sprites['P'] = example.PlayerSprite
  return ascii_art.ascii_art_to_game(art, ' ', sprites, {'X': example.JudgeDrape},
                                     update_schedule=[boxes, ['X'], ['P']])


# ---- an extraterrestrial_marauders board that the reference does not ship -----------
# Same entity set (the example's classes read the module's bolt character lists),
# on a 14x27 board: rows of one 32-bit word instead of two, three marauder rows,
# three bunkers.  Row 10 stays the invasion line (extraterrestrial_marauders.py:147).
MARAUDERS_ART = {
    'marauders_custom_A': ['   X  X  X  X  X  X  X     ',
                           '    X  X  X  X  X  X  X    ',
                           '   X  X  X  X  X  X  X     ',
                           '                           ',
                           '                           ',
                           '                           ',
                           '                           ',
                           '                           ',
                           '                           ',
                           '                           ',
                           '   BBB     BBB     BBB     ',
                           '   BBB     BBB     BBB     ',
                           '                           ',
                           '  P                        '],
}
MARAUDERS_NAMES = sorted(MARAUDERS_ART)
for _name, _art in MARAUDERS_ART.items():
  assert len(set(len(_row) for _row in _art)) == 1, _name


def make_marauders(name, example, ascii_art):
  """extraterrestrial_marauders.make_game (:91-101) for MARAUDERS_ART[name]."""
  bolts = example.UPWARD_BOLT_CHARS + example.DOWNWARD_BOLT_CHARS
  return ascii_art.ascii_art_to_game(
      MARAUDERS_ART[name], what_lies_beneath=' ',
      sprites=dict([('P', example.PlayerSprite)] +
                   [(c, example.UpwardLaserBoltSprite) for c in example.UPWARD_BOLT_CHARS] +
                   [(c, example.DownwardLaserBoltSprite) for c in example.DOWNWARD_BOLT_CHARS]),
      drapes=dict(X=example.MarauderDrape, B=example.BunkerDrape),
      update_schedule=['P', 'B', 'X'] + list(bolts))


# ---- a hello_world board that the reference does not ship ---------------------------
# 8x33: curtain rows of 33 bits (one bit into the second word), another z-order,
# sprites with the four direction sets in another order.
HELLO_ART = {
    'hello_custom_A': ['                                 ',
                       ' @@@  @   @   1     #   #        ',
                       ' @  @ @@ @@         ## ##     2  ',
                       ' @@@  @ @ @    3    # # #        ',
                       ' @    @   @         #   #    @@@@',
                       ' @    @   @  4      #   #        ',
                       '@                               @',
                       '                                 '],
}
HELLO_NAMES = sorted(HELLO_ART)
for _name, _art in HELLO_ART.items():
  assert len(set(len(_row) for _row in _art)) == 1, _name


def make_hello(name, example, ascii_art):
  """hello_world.make_game (hello_world.py:59-69) for HELLO_ART[name]."""
  return ascii_art.ascii_art_to_game(
      HELLO_ART[name], what_lies_beneath=' ',
      sprites={'1': ascii_art.Partial(example.SlidingSprite, 3),
               '2': ascii_art.Partial(example.SlidingSprite, 0),
               '3': ascii_art.Partial(example.SlidingSprite, 2),
               '4': ascii_art.Partial(example.SlidingSprite, 1)},
      drapes={'@': example.RollingDrape},
      z_order='4@321')


# ---- better_scrolly_maze boards that the reference does not ship ---------------------
# The example's own classes (PlayerSprite, PatrollerSprite, CashDrape;
# better_scrolly_maze.py:250-320) on boards without a compiled instance of
# pcx_better_scrolly_step (its run-time-shape instance): 17x38 (646 cells: the
# planes are padded to whole dwords) and 12x20.
# name -> (seed, rows, cols)
BETTER_SPECS = {
    'better_scrolly_custom_A': (201, 17, 38),
    'better_scrolly_custom_B': (202, 12, 20),
    # (seed, rows, cols, coins kept): a small board with only the two coins nearest the player, so that tapes collect
    # them ALL -- "no coins left ends the episode" (better_scrolly_maze.py:317-320) is pinned by no other fixture
    # (oracle/mutants.py: `better_last_coin_does_not_end_the_episode` survived)
    'better_scrolly_custom_C': (203, 9, 14, 2),
}
BETTER_NAMES = sorted(BETTER_SPECS)


# An UNWALLED board: everybody can walk off it, and a thing off the board has position (0, 0) (sprites.py:391-411) --
# which is where better_scrolly_maze's kill test (TRUE positions, better_scrolly_maze.py:300) and coin pickup (:313)
# then happen: patroller `a` leaves the board to the west every lap (row 1 has no wall in its last column, which is
# where `layers['#'][row, -1]` looks), turns on the wall it then "sees" at [0, -1], and kills a player who stands at
# (0, 0) or is off the board too; the coin at (0, 0) is collected from anywhere outside.  `b` turns in column 0 on the
# wall numpy's index -1 finds in the LAST column.  No patroller can reach the last column (the IndexError of
# `col + 1`), so tapes never raise.  (oracle/mutants.py: `coin_taken_at_the_virtual_position` survived every other fixture.)
BETTER_ART = {
    'better_scrolly_custom_D': ['@          #',
                                '  a     #   ',
                                '      P     ',
                                '   b      ##',
                                '   @    @   ',
                                ' #      c # ',
                                '@           '],
    # Walls on BOTH sides of board cell (0, 0) -- [0, 1] and, through numpy's index -1, [0, -1]: patrollers `a` (east to the
    # wall in column 8, then west) and `c` (east to column 9, then west) leave the board in column 0 of rows whose last
    # column is free, and out there they look around position (0, 0): `if layers['#'][row, col-1]: east` THEN
    # `if layers['#'][row, col+1]: west` (better_scrolly_maze.py:291-294) -- the second test wins, they walk west for good.
    # A restatement that lets a walled-in patroller turn the other way (oracle/mutants.py) brings them back on the board.
    # `b` turns in column 0 on the wall index -1 finds in the last column, as in custom_D.  Round 6 (VERDICT r5 missing #3c).
    'better_scrolly_custom_E': ['@#         #',
                                '    a   #   ',
                                '      P     ',
                                '   b      ##',
                                '   @    @   ',
                                '        c # ',
                                '@           '],
}
BETTER_NAMES = BETTER_NAMES + sorted(BETTER_ART)


def better_art(name):
  if name in BETTER_ART:
    return BETTER_ART[name]
  seed, rows, cols = BETTER_SPECS[name][:3]
  rng = np.random.RandomState(seed)
  art = np.full((rows, cols), ' ', dtype='<U1')
  art[0, :] = art[-1, :] = art[:, 0] = art[:, -1] = '#'
  inner = rng.rand(rows - 2, cols - 2)
  art[1:-1, 1:-1][inner < 0.18] = '#'
  art[1:-1, 1:-1][(inner >= 0.18) & (inner < 0.26)] = '@'
  for ch in 'abcP':
    for _ in range(10000):
      r = int(rng.randint(1, rows - 1)); c = int(rng.randint(1, cols - 1))
      if art[r, c] == ' ':
        art[r, c] = ch
        break
    else:
      raise RuntimeError('no room for sprite ' + ch)
  if len(BETTER_SPECS[name]) > 3:  # only the coins nearest the player stay
    (pr,), (pc,) = np.nonzero(art == 'P')
    coins = sorted(zip(*np.nonzero(art == '@')), key=lambda rc: (abs(rc[0] - pr) + abs(rc[1] - pc), rc))
    for r, c in coins[BETTER_SPECS[name][3]:]:
      art[r, c] = ' '
  return [''.join(row) for row in art]


def make_better_scrolly(name, example, ascii_art):
  """better_scrolly_maze.make_game (better_scrolly_maze.py:209-222) for BETTER_SPECS[name]."""
  return ascii_art.ascii_art_to_game(
      better_art(name), what_lies_beneath=' ',
      sprites={'P': example.PlayerSprite, 'a': example.PatrollerSprite, 'b': example.PatrollerSprite,
               'c': example.PatrollerSprite},
      drapes={'@': example.CashDrape},
      update_schedule=['a', 'b', 'c', 'P', '@'], z_order='abc@P')
