"""`MazeWalker`: the host-side template of a maze-walking sprite.

API surface of the reference's `prefab_parts/sprites.py:27-575`.  The
constructor and `_teleport` run on the host (game files call them while the
game is being built); the motion helpers `_north` ... `_stay` are what a
subclass' `update()` would call, and those run on the device
(`mw_move` / `check_motion` / `teleport` in `csrc/pcx_scrolly_maze.hip`,
`csrc/pcx_generic.hip`, `csrc/pcx_warehouse.hip`, `csrc/pcx_marauders.hip`).
"""

from pycolab_amd import things


def _require_ascii_items(items, argument_name, function_name):
  for item in items:
    try:
      ord(item)
    except TypeError:
      raise TypeError(
          '{} requires all elements in its {} argument to be single-character '
          'ASCII strings, but {} was found inside {}.'.format(
              function_name, argument_name, repr(item), argument_name))


class DeviceOnlyError(NotImplementedError):
  """Raised when step-time entity logic is invoked on the host."""


def _device_only(name):
  def method(self, *unused_args, **unused_kwargs):
    raise DeviceOnlyError(
        '{}.{}() is step-time logic: it runs inside the HIP step kernel, not '
        'on the host'.format(type(self).__name__, name))
  method.__name__ = name
  return method


class MazeWalker(things.Sprite):
  """Sprite that moves in the 8 grid directions and respects obstacles."""

  EDGE = 'edge!'

  _NORTH = (-1, 0)
  _NORTHEAST = (-1, 1)
  _EAST = (0, 1)
  _SOUTHEAST = (1, 1)
  _SOUTH = (1, 0)
  _SOUTHWEST = (1, -1)
  _WEST = (0, -1)
  _NORTHWEST = (-1, -1)
  _STAY = (0, 0)

  def __init__(self, corner, position, character, impassable,
               confined_to_board=False,
               egocentric_scroller=False,
               scrolling_group=''):
    super(MazeWalker, self).__init__(corner, position, character)
    _require_ascii_items(impassable, 'impassable', 'the MazeWalker constructor')
    if character in impassable:
      raise ValueError('A MazeWalker must not designate its own character {} '
                       'as impassable.'.format(repr(character)))
    self._impassable = set(impassable)
    self._confined_to_board = confined_to_board
    self._egocentric_scroller = egocentric_scroller
    self._scrolling_group = scrolling_group
    self._virtual_row, self._virtual_col = position
    self._prior_visible = None

  @property
  def virtual_position(self):
    return self.Position(self._virtual_row, self._virtual_col)

  @property
  def on_the_board(self):
    return self._on_board(self._virtual_row, self._virtual_col)

  @property
  def impassable(self):
    return self._impassable

  def _on_board(self, row, col):
    return (0 <= row < self.corner.row) and (0 <= col < self.corner.col)

  def _on_board_exit(self):
    self._prior_visible = self._visible
    self._visible = False

  def _on_board_enter(self):
    self._visible = self._prior_visible

  def _teleport(self, virtual_position):
    """Host restatement of sprites.py:315-352 (used by constructors)."""
    row, col = virtual_position
    was_on = self._on_board(self._virtual_row, self._virtual_col)
    now_on = self._on_board(row, col)
    if was_on and not now_on:
      self._on_board_exit()
    self._virtual_row, self._virtual_col = row, col
    self._position = self.Position(row, col) if now_on else self.Position(0, 0)
    if now_on and not was_on:
      self._on_board_enter()

  _northwest = _device_only('_northwest')
  _north = _device_only('_north')
  _northeast = _device_only('_northeast')
  _east = _device_only('_east')
  _southeast = _device_only('_southeast')
  _south = _device_only('_south')
  _southwest = _device_only('_southwest')
  _west = _device_only('_west')
  _stay = _device_only('_stay')
