"""Scrolling-protocol constants (reference: protocols/scrolling.py:255-285).

On the device the whole protocol collapses to a few per-environment scalars
(current order + its frame stamp, a registered bit, a 9-bit permit mask + its
frame stamp per egocentric sprite, per scrolling group); see `mw_move` /
`maybe_move` / `is_possible` in csrc/pcx_scrolly_maze.hip and
csrc/pcx_generic.hip.  The host only needs
the motion names and the exception type.
"""

NORTH = (-1, 0)
NORTHEAST = (-1, 1)
EAST = (0, 1)
SOUTHEAST = (1, 1)
SOUTH = (1, 0)
SOUTHWEST = (1, -1)
WEST = (0, -1)
NORTHWEST = (-1, -1)


class Error(RuntimeError):
  """Mishandling of the scrolling protocol (scrolling.py:279)."""
