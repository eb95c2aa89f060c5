"""Host-side entity classes: `Backdrop`, `Drape`, `Sprite`.

These keep the constructor and property surface of the reference's
`pycolab/things.py:57-391` so that game files written against pycolab import
and construct unchanged.  In this framework an entity object is a *template*:
its constructor runs once on the host, the template compiler
(`pycolab_amd.compiler`) turns the constructed object into plain data, and the
per-step `update()` logic runs as a hand-written HIP device program selected
by class (`pycolab_amd.programs`).  `update()` bodies are therefore never
called by this package.
"""

import collections


class Backdrop(object):
  """Background scenery (reference: things.py:57-158)."""

  def __init__(self, curtain, palette):
    self.__curtain = curtain
    self.__palette = palette

  def update(self, actions, board, layers, things, the_plot):
    """The base backdrop never changes (things.py:146-147)."""

  @property
  def curtain(self):
    return self.__curtain

  @property
  def palette(self):
    return self.__palette


class Drape(object):
  """A binary mask painted with one character (reference: things.py:161-247)."""

  def __init__(self, curtain, character):
    self.__curtain = curtain
    self.__character = character

  def update(self, actions, board, layers, backdrop, things, the_plot):
    raise NotImplementedError(
        'Drape.update() runs as a device program in pycolab_amd; '
        'see pycolab_amd.programs')

  @property
  def character(self):
    return self.__character

  @property
  def curtain(self):
    return self.__curtain


class Sprite(object):
  """A single-cell entity (reference: things.py:250-391)."""

  Position = collections.namedtuple('Position', ['row', 'col'])

  def __init__(self, corner, position, character):
    self.__corner = corner
    self.__character = character
    # The two members below are the ones subclasses are allowed to touch
    # (things.py:316-319).
    self._position = position
    self._visible = True

  def update(self, actions, board, layers, backdrop, things, the_plot):
    raise NotImplementedError(
        'Sprite.update() runs as a device program in pycolab_amd; '
        'see pycolab_amd.programs')

  @property
  def character(self):
    return self.__character

  @property
  def corner(self):
    return self.__corner

  @property
  def position(self):
    return self._position

  @property
  def visible(self):
    return self._visible
