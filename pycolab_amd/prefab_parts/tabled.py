"""Prefab entities whose whole `update()` is "map the action to a motion".

`TabledWalker` is a `MazeWalker` and `TabledScrolly` a `Scrolly` with the
update rule of the reference's test entities (tests/test_things.py:203-295):
the action selects one of the nine motions and the matching motion helper
(`_north` ... `_stay`) is called.  With integer actions: 0 N, 1 NE, 2 E, 3 SE,
4 S, 5 SW, 6 W, 7 NW; anything else (`None` included) is `_stay`.

Several independently controlled entities share one integer action by
packing: an entity built with `action_field=(shift, mask)` obeys
`(action >> shift) & mask`.  Games made only of these (plus static things)
need no hand-written device program: the table-driven kernel
(csrc/pcx_generic.hip) runs them as they are.
"""

from pycolab_amd.prefab_parts import drapes
from pycolab_amd.prefab_parts import sprites

MOTIONS = ['n', 'ne', 'e', 'se', 's', 'sw', 'w', 'nw', 'stay']


class TabledWalker(sprites.MazeWalker):
  pcx_program = 'walker'

  def __init__(self, corner, position, character, impassable,
               confined_to_board=False, egocentric_scroller=False,
               scrolling_group='', action_field=(0, 0)):
    super(TabledWalker, self).__init__(
        corner, position, character, impassable, confined_to_board,
        egocentric_scroller, scrolling_group)
    self.pcx_param = (int(action_field[0]), int(action_field[1]), 0, 0)


class TabledScrolly(drapes.Scrolly):
  pcx_program = 'scrolly'

  def __init__(self, curtain, character, board_shape, whole_pattern,
               board_northwest_corner, scroll_margins=(2, 3),
               scrolling_group='', action_field=(0, 0)):
    super(TabledScrolly, self).__init__(
        curtain, character, board_shape, whole_pattern,
        board_northwest_corner, scroll_margins, scrolling_group)
    self.pcx_param = (int(action_field[0]), int(action_field[1]), 0, 0)


class StaticDrape(drapes.things.Drape):
  """A drape that never changes."""
  pcx_program = 'static'
