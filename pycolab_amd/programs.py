"""Registry: which hand-written device program implements an entity class.

The reference runs arbitrary Python in `update()`; the HIP engine runs a
device program per entity instead.  A class is matched either

* explicitly: the class (or a base) carries `pcx_program = '<name>'`, or
* by identity with a shipped reference game class: same class name and the
  same `update()` bytecode fingerprint as the class in the reference's
  `examples/*.py` (so an edited copy is *not* silently mis-simulated).

Anything else raises `UnsupportedEntityError` -- there is no CPU fallback.
"""

import hashlib

from pycolab_amd import _native as N
from pycolab_amd import things

NAMES = {
    'scrolly_maze.player': N.PROG_SM_PLAYER,
    'scrolly_maze.patroller': N.PROG_SM_PATROLLER,
    'scrolly_maze.maze': N.PROG_SM_MAZE,
    'scrolly_maze.cash': N.PROG_SM_CASH,
    'marauders.player': N.PROG_EM_PLAYER,
    'marauders.bunker': N.PROG_EM_BUNKER,
    'marauders.marauder': N.PROG_EM_MARAUDER,
    'marauders.upward_bolt': N.PROG_EM_UPBOLT,
    'marauders.downward_bolt': N.PROG_EM_DOWNBOLT,
    'warehouse.box': N.PROG_WM_BOX,
    'warehouse.judge': N.PROG_WM_JUDGE,
    'warehouse.player': N.PROG_WM_PLAYER,
    'hello_world.rolling': N.PROG_HW_ROLLING,
    'hello_world.sliding': N.PROG_HW_SLIDING,
    'better_scrolly_maze.player': N.PROG_BS_PLAYER,
    'better_scrolly_maze.patroller': N.PROG_BS_PATROLLER,
    'better_scrolly_maze.cash': N.PROG_BS_CASH,
    'walker': N.PROG_WALKER,
    'scrolly': N.PROG_SCROLLY,
    'static': N.PROG_STATIC,
}

GAME_OF_PROGRAM = {
    N.PROG_SM_PLAYER: N.GAME_SCROLLY_MAZE, N.PROG_SM_PATROLLER: N.GAME_SCROLLY_MAZE,
    N.PROG_SM_MAZE: N.GAME_SCROLLY_MAZE, N.PROG_SM_CASH: N.GAME_SCROLLY_MAZE,
    N.PROG_EM_PLAYER: N.GAME_MARAUDERS, N.PROG_EM_BUNKER: N.GAME_MARAUDERS,
    N.PROG_EM_MARAUDER: N.GAME_MARAUDERS, N.PROG_EM_UPBOLT: N.GAME_MARAUDERS,
    N.PROG_EM_DOWNBOLT: N.GAME_MARAUDERS,
    N.PROG_WM_BOX: N.GAME_WAREHOUSE, N.PROG_WM_JUDGE: N.GAME_WAREHOUSE,
    N.PROG_WM_PLAYER: N.GAME_WAREHOUSE,
    N.PROG_HW_ROLLING: N.GAME_HELLO_WORLD, N.PROG_HW_SLIDING: N.GAME_HELLO_WORLD,
    N.PROG_BS_PLAYER: N.GAME_BETTER_SCROLLY, N.PROG_BS_PATROLLER: N.GAME_BETTER_SCROLLY,
    N.PROG_BS_CASH: N.GAME_BETTER_SCROLLY,
}

# Number of "ordinary" actions per game (quit excluded): SURVEY.md section 8(d).
N_ACTIONS = {N.GAME_SCROLLY_MAZE: 5, N.GAME_MARAUDERS: 4, N.GAME_WAREHOUSE: 5,
             N.GAME_HELLO_WORLD: 4, N.GAME_WALKERS: 9, N.GAME_BETTER_SCROLLY: 5}

# (class name, fingerprint of update()) of the reference's shipped game
# classes -> program name.  Fingerprints are produced by
# `oracle/gen_templates.py --fingerprints` from /root/reference on the pinned
# interpreter (CPython 3.10).
SHIPPED = {
    # filled in by oracle/gen_templates.py; see SHIPPED_FINGERPRINTS below
}


def _pack2(values):
  """Four values in {-1, 0, 1} as 2-bit fields (value + 1)."""
  return sum(((int(v) + 1) & 3) << (2 * i) for i, v in enumerate(values))


# Per-program constants that the entity's constructor left on the object
# (reference attribute names), delivered to the device program as param[0..3].
PARAM_EXTRACTORS = {
    N.PROG_SM_PATROLLER: lambda e: [int(bool(e._moving_east)), 0, 0, 0],     # scrolly_maze.py:282
    N.PROG_BS_PATROLLER: lambda e: [int(bool(e._moving_east)), 0, 0, 0],     # better_scrolly_maze.py:282
    N.PROG_HW_SLIDING: lambda e: [_pack2(e._dx), _pack2(e._dy), 0, 0],      # hello_world.py:114-115
    N.PROG_EM_MARAUDER: lambda e: [int(e._dx), 0, 0, 0],                    # marauders.py:139
    N.PROG_WM_JUDGE: lambda e: [int(e._last_num_boxes_on_goals), 0, 0, 0],  # warehouse_manager.py:243
}


def extract_params(entity, program):
  if program in PARAM_EXTRACTORS:
    return PARAM_EXTRACTORS[program](entity)
  return [int(v) for v in getattr(entity, 'pcx_param', (0, 0, 0, 0))]


class UnsupportedEntityError(NotImplementedError):
  pass


def _digest_code(h, code):
  """Feeds what identifies a code object's behaviour -- not where it lives in
  memory or how a set constant happens to be ordered in this process."""
  h.update(code.co_code)
  h.update(repr(code.co_names).encode())
  for c in code.co_consts:
    if hasattr(c, 'co_code'):           # nested lambda / comprehension
      h.update(b'<code>')
      _digest_code(h, c)
    elif isinstance(c, frozenset):      # `x in {..}` constants: iteration order is per-process
      h.update(repr(sorted(repr(e) for e in c)).encode())
    elif isinstance(c, str) and len(c) >= 40:
      continue                          # docstrings
    else:
      h.update(repr(c).encode())


def fingerprint(cls):
  """Stable digest of the bytecode of every method the class itself defines
  (same interpreter version only; identical across processes)."""
  h = hashlib.sha1()
  found = False
  for name in sorted(vars(cls)):
    code = getattr(vars(cls)[name], '__code__', None)
    if code is None:
      continue
    found = True
    h.update(name.encode())
    _digest_code(h, code)
  return h.hexdigest()[:16] if found else None


def resolve(entity):
  """Device program id for a constructed Sprite/Drape/Backdrop object."""
  cls = type(entity)
  named = getattr(cls, 'pcx_program', None)
  if named is not None:
    if named not in NAMES:
      raise UnsupportedEntityError(
          '{}.pcx_program = {!r} names no device program'.format(cls.__name__, named))
    return NAMES[named]
  key = (cls.__name__, fingerprint(cls))
  if key in SHIPPED:
    return NAMES[SHIPPED[key]]
  if isinstance(entity, things.Backdrop):
    if cls.update is things.Backdrop.update:
      return N.PROG_STATIC
  raise UnsupportedEntityError(
      'No device program is registered for entity class {}.{} (update() '
      'fingerprint {}). pycolab_amd steps games on the GPU only; give the '
      'class a `pcx_program` attribute naming one of {} or add a program to '
      'pycolab_amd/csrc.'.format(cls.__module__, cls.__name__, key[1],
                                 sorted(NAMES)))


def infer_game(program_ids):
  games = {GAME_OF_PROGRAM[p] for p in program_ids if p in GAME_OF_PROGRAM}
  if len(games) > 1:
    raise UnsupportedEntityError(
        'entities of different shipped games cannot be mixed in one engine')
  if games:
    return games.pop()
  return N.GAME_WALKERS


try:
  from pycolab_amd._shipped_fingerprints import SHIPPED_FINGERPRINTS
  SHIPPED.update(SHIPPED_FINGERPRINTS)
except ImportError:  # pragma: no cover - file is committed
  pass
