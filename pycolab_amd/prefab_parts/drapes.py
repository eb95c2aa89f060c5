"""`Scrolly`: the host-side template of a scrolling drape, and `PatternInfo`.

API surface of the reference's `prefab_parts/drapes.py:30-695`.  Constructors
run on the host while a game is built; `_maybe_move` and the scrolling
protocol run on the device (`maybe_move` in `csrc/pcx_scrolly_maze.hip` and
`csrc/pcx_generic.hip`).
"""

import numpy as np

from pycolab_amd import ascii_art
from pycolab_amd import things
from pycolab_amd.prefab_parts.sprites import _device_only


class Scrolly(things.Drape):
  """Drape showing a board-sized window onto a larger binary pattern."""

  _NORTH = (-1, 0)
  _NORTHEAST = (-1, 1)
  _EAST = (0, 1)
  _SOUTHEAST = (1, 1)
  _SOUTH = (1, 0)
  _SOUTHWEST = (1, -1)
  _WEST = (0, -1)
  _NORTHWEST = (-1, -1)
  _STAY = (0, 0)

  class PatternInfo(object):
    """Derives `Scrolly` constructor arguments from world ASCII art
    (drapes.py:166-291)."""

    def __init__(self, whole_pattern_art, board_art_or_shape,
                 board_northwest_corner_mark, what_lies_beneath):
      if ord(what_lies_beneath) > 127:
        raise ValueError(
            'The what_lies_beneath value used to build a Scrolly.PatternInfo '
            'must be an ASCII character.')
      self._art = ascii_art.ascii_art_to_uint8_nparray(whole_pattern_art)
      self._corner = self._locate(
          board_northwest_corner_mark, 'the Scrolly.PatternInfo constructor')
      self._art[self._corner] = ord(what_lies_beneath)
      try:
        self._board_shape = (len(board_art_or_shape),
                             len(board_art_or_shape[0]))
      except TypeError:
        rows, cols = board_art_or_shape
        self._board_shape = (rows, cols)
      if (self._board_shape[0] > self._art.shape[0] or
          self._board_shape[1] > self._art.shape[1]):
        raise ValueError(
            'The whole_pattern_art value used to build a Scrolly.PatternInfo '
            '(size {}) cannot completely cover the game board (size '
            '{}).'.format(self._art.shape, self._board_shape))

    def virtual_position(self, character):
      where = self._locate(character, 'Scrolly.PatternInfo.virtual_position()')
      return (where[0] - self._corner[0], where[1] - self._corner[1])

    def kwargs(self, character):
      return {'board_shape': self._board_shape,
              'whole_pattern': self._art == ord(character),
              'board_northwest_corner': self._corner}

    def _locate(self, character, who):
      hits = np.argwhere(self._art == ord(character))
      if len(hits) == 0:
        raise RuntimeError(
            '{} found no instances of {} in the pattern art used to build this '
            'PatternInfo object.'.format(who, repr(character)))
      if len(hits) > 1:
        raise RuntimeError(
            '{} found multiple instances of {} in the pattern art used to '
            'build this PatternInfo object.'.format(who, repr(character)))
      return (int(hits[0][0]), int(hits[0][1]))

  def __init__(self, curtain, character, board_shape,
               whole_pattern, board_northwest_corner,
               scroll_margins=(2, 3), scrolling_group=''):
    super(Scrolly, self).__init__(curtain, character)
    self._board_shape = board_shape
    self._northwest_corner = board_northwest_corner
    self._scrolling_group = scrolling_group
    self.__whole_pattern = whole_pattern
    self._northwest_corner_limit = (whole_pattern.shape[0] - board_shape[0],
                                    whole_pattern.shape[1] - board_shape[1])
    if min(self._northwest_corner_limit) < 0:
      raise ValueError(
          'The whole_pattern provided to the `Scrolly` constructor (size {}) '
          'cannot completely cover the game board (size {}).'.format(
              whole_pattern.shape, board_shape))
    self._have_margins = scroll_margins is not None
    self._scroll_margins = scroll_margins
    if self._have_margins:
      # drapes.py:355-358
      self._margin_north = scroll_margins[0] - 1
      self._margin_south = board_shape[0] - scroll_margins[0]
      self._margin_west = scroll_margins[1] - 1
      self._margin_east = board_shape[1] - scroll_margins[1]
      if (self._margin_west >= self._margin_east or
          self._margin_north >= self._margin_south):
        raise ValueError(
            'The scrolling margins provided to the `Scrolly` constructor, {}, '
            'are so large that a margin would overlap more than half of the '
            'board.'.format(scroll_margins))
    self._update_curtain()
    self._last_maybe_move_frame = -float('inf')
    self._prescroll_northwest_corner = self._northwest_corner

  @property
  def whole_pattern(self):
    return self.__whole_pattern

  def _update_curtain(self):
    r0, c0 = self._northwest_corner
    np.copyto(self.curtain,
              self.whole_pattern[r0:r0 + self._board_shape[0],
                                 c0:c0 + self._board_shape[1]])

  pattern_position_prescroll = _device_only('pattern_position_prescroll')
  pattern_position_postscroll = _device_only('pattern_position_postscroll')
  _northwest = _device_only('_northwest')
  _north = _device_only('_north')
  _northeast = _device_only('_northeast')
  _east = _device_only('_east')
  _southeast = _device_only('_southeast')
  _south = _device_only('_south')
  _southwest = _device_only('_southwest')
  _west = _device_only('_west')
  _stay = _device_only('_stay')
