"""Registry: which hand-written device program implements an entity class.

The reference runs arbitrary Python in `update()`; the HIP engine runs a
device program per entity instead.  A class is matched either

* explicitly: the class (or a base) carries `pcx_program = '<name>'`, or
* by identity with a shipped reference game class: same class name and the
  same source (digest of the normalised AST of the class body: comments,
  formatting, docstrings and the interpreter version do not matter) as the
  class in the reference's `examples/*.py` -- so an edited copy is *not*
  silently mis-simulated.

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
    'ordeal.player': N.PROG_OD_PLAYER,
    'ordeal.dragonduck': N.PROG_OD_DRAGONDUCK,
    'ordeal.sword': N.PROG_OD_SWORD,
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

# examples/ordeal.py: the programs of its three classes run in the table-driven kernel (GAME_WALKERS); they keep Plot
# entries in the engine's plot words, add float rewards, and name the Story's chapters by code = index among these keys
# (include/pcx.h PCX_PROG_OD_PLAYER)
ORDEAL_PROGRAMS = (N.PROG_OD_PLAYER, N.PROG_OD_DRAGONDUCK, N.PROG_OD_SWORD)
ORDEAL_CHAPTERS = ('castle', 'cavern', 'kansas')
ORDEAL_N_ACTIONS = 5  # ordeal.py:216-246: 0 N, 1 S, 2 W, 3 E, 4 quit

# Number of "ordinary" actions per game (quit excluded): SURVEY.md section 8(d).
N_ACTIONS = {N.GAME_SCROLLY_MAZE: 5, N.GAME_MARAUDERS: 4, N.GAME_WAREHOUSE: 5,
             N.GAME_HELLO_WORLD: 4, N.GAME_WALKERS: 9, N.GAME_BETTER_SCROLLY: 5}

# (class name, source fingerprint) of the reference's shipped game classes ->
# program name.  Produced by `oracle/gen_templates.py` from /root/reference;
# independent of the interpreter that runs it.
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


def _canonical(node, out):
  """Feeds a version-independent serialisation of an AST into `out`: node
  type names and field values only -- no line numbers, no docstrings, and the
  node kinds that older interpreters spell differently folded together
  (Num/Str/Bytes/NameConstant -> Constant, Index unwrapped), so CPython 3.8
  through 3.13 agree on the digest of the same source text."""
  import ast
  if isinstance(node, ast.AST):
    name = type(node).__name__
    if name == 'Index':                      # py < 3.9: Subscript(slice=Index(value))
      return _canonical(node.value, out)
    if name in ('Num', 'Str', 'Bytes', 'NameConstant', 'Ellipsis'):  # py < 3.8 spellings
      value = getattr(node, 'n', getattr(node, 's', getattr(node, 'value', Ellipsis)))
      out.append('Constant(%r)' % (value,))
      return
    if name == 'Constant':
      out.append('Constant(%r)' % (node.value,))
      return
    out.append(name + '(')
    for field in node._fields:
      if field in ('ctx', 'type_comment', 'kind', 'type_ignores', 'type_params'):
        continue
      value = getattr(node, field, None)
      if field == 'body' and isinstance(value, list) and value and name in ('FunctionDef', 'ClassDef', 'Module'):
        first = value[0]  # drop the docstring
        if isinstance(first, ast.Expr) and type(first.value).__name__ in ('Constant', 'Str') and isinstance(
            getattr(first.value, 'value', getattr(first.value, 's', None)), str):
          value = value[1:] or [ast.Pass()]
      out.append(field + '=')
      _canonical(value, out)
    out.append(')')
  elif isinstance(node, list):
    out.append('[')
    for item in node:
      _canonical(item, out)
      out.append(',')
    out.append(']')
  else:
    out.append(repr(node))


def fingerprint(cls):
  """Digest of the class's own source text, as a normalised AST (comments,
  formatting, docstrings and the interpreter version do not matter; any change
  to the code does).  None when the source is not available."""
  import ast
  import inspect
  import textwrap
  node = None
  try:
    tree = ast.parse(textwrap.dedent(inspect.getsource(cls)))
    node = tree.body[0] if tree.body else None
  except (OSError, TypeError, SyntaxError, IndentationError):
    # a module loaded by path and never entered into sys.modules: find the
    # file through one of the class's own functions and pick the ClassDef
    for member in vars(cls).values():
      code = getattr(member, '__code__', None)
      if code is None:
        continue
      try:
        with open(code.co_filename, 'rb') as f:
          tree = ast.parse(f.read())
      except (OSError, SyntaxError, ValueError):
        break
      scope = tree.body
      for part in cls.__qualname__.split('.'):
        if part == '<locals>':
          continue
        found = [n for n in scope if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name == part]
        if not found:
          scope = None
          break
        node, scope = found[-1], found[-1].body
      if scope is None:
        node = None
      break
  if not isinstance(node, ast.ClassDef):
    return None
  out = []
  _canonical(node, out)
  return hashlib.sha1(''.join(out).encode('utf-8')).hexdigest()[:16]


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
      'No device program is registered for entity class {}.{} (source '
      'fingerprint {}: it is not one of the reference\'s shipped example classes, or '
      'it has been edited). pycolab_amd steps games on the GPU only; give the '
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
