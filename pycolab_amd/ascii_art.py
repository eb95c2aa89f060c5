"""Build games from ASCII-art diagrams.

Same public surface as the reference's `pycolab/ascii_art.py:31-364`
(`ascii_art_to_game`, `ascii_art_to_uint8_nparray`, `Partial`).  This is the
host-side boundary of the hot path: it runs once per template, constructs the
user's entity objects, and leaves an `Engine` whose `its_showtime()` compiles
the template to plain data and hands it to the HIP engine.
"""

import itertools

import numpy as np

from pycolab_amd import things


def ascii_art_to_uint8_nparray(art):
  """Stack equal-length ASCII strings into a uint8 array (ascii_art.py:295-328)."""
  complaint = (
      'the argument to ascii_art_to_uint8_nparray must be a list (or tuple) '
      'of strings containing the same number of strictly-ASCII characters.')
  try:
    rows = [np.frombuffer(line.encode('ascii'), dtype=np.uint8) for line in art]
    out = np.vstack(rows)
  except AttributeError as e:
    if isinstance(art, (list, tuple)) and all(
        isinstance(row, (list, tuple)) for row in art):
      complaint += ' Did you pass a list of list of single characters?'
    raise TypeError('{} (original error: {})'.format(complaint, e))
  except ValueError as e:
    raise ValueError('{} (original error from numpy: {})'.format(complaint, e))
  if np.any(out > 127):
    raise ValueError(complaint)
  return out


class Partial(object):
  """An entity class plus the extra constructor arguments it should get
  (ascii_art.py:331-364)."""

  def __init__(self, pycolab_thing, *args, **kwargs):
    if not issubclass(pycolab_thing,
                      (things.Backdrop, things.Sprite, things.Drape)):
      raise TypeError('the pycolab_thing argument to ascii_art.Partial must be '
                      'a Backdrop, Sprite, or Drape subclass.')
    self.pycolab_thing = pycolab_thing
    self.args = args
    self.kwargs = kwargs


def _as_partial(value):
  return value if isinstance(value, Partial) else Partial(value)


def ascii_art_to_game(art,
                      what_lies_beneath,
                      sprites=None, drapes=None, backdrop=things.Backdrop,
                      update_schedule=None,
                      z_order=None,
                      occlusion_in_layers=True):
  """Construct an `Engine` from ASCII art (ascii_art.py:31-291).

  Arguments, validation order, error types and messages follow the reference;
  see its docstring for the full description.  The returned engine is a
  template until `its_showtime()`; call `engine.configure(batch=..., device=...)`
  first to step many environments at once.
  """
  from pycolab_amd import engine  # late: engine imports this module's helpers

  sprites = {c: _as_partial(v) for c, v in (sprites or {}).items()}
  drapes = {c: _as_partial(v) for c, v in (drapes or {}).items()}
  backdrop = _as_partial(backdrop)

  movers = set(sprites) | set(drapes)
  if update_schedule is None:
    update_schedule = list(movers)
  if isinstance(update_schedule, str):
    update_schedule = list(update_schedule)
  if all(isinstance(item, str) for item in update_schedule):
    update_schedule = [update_schedule]
  try:
    flat_schedule = list(itertools.chain.from_iterable(update_schedule))
  except TypeError:
    raise TypeError('if any element in update_schedule is an iterable (like a '
                    'list), all elements in update_schedule must be')
  if set(flat_schedule) != movers:
    raise ValueError('if specified, update_schedule must list each sprite and '
                     'drape exactly once.')

  if z_order is None:
    z_order = flat_schedule
  if set(z_order) != movers:
    raise ValueError('if specified, z_order must list each sprite and drape '
                     'exactly once.')

  if isinstance(what_lies_beneath, str) and len(what_lies_beneath) != 1:
    raise ValueError(
        'what_lies_beneath may either be a single-character ASCII string or '
        'a list of ASCII-character strings')
  try:
    for group in (''.join(what_lies_beneath), movers, z_order, flat_schedule):
      for character in group:
        ord(character)
  except TypeError:
    raise ValueError(
        'keys of sprites, keys of drapes, what_lies_beneath (or its entries), '
        'values in z_order, and (possibly nested) values in update_schedule '
        'must all be single-character ASCII strings.')
  if movers.intersection(''.join(what_lies_beneath)):
    raise ValueError(
        'any character specified in what_lies_beneath must not be one of the '
        'characters used as keys in the sprites or drapes arguments.')

  art = ascii_art_to_uint8_nparray(art)
  if isinstance(what_lies_beneath, str):
    beneath = np.full_like(art, ord(what_lies_beneath))
  else:
    beneath = ascii_art_to_uint8_nparray(what_lies_beneath)
    if art.shape != beneath.shape:
      raise ValueError(
          'if not a single ASCII character, what_lies_beneath must be ASCII '
          'art whose shape is the same as that of the ASCII art in art.')

  group_name = {}
  for index, group in enumerate(update_schedule):
    for character in group:
      group_name[character] = '{:05d}'.format(index)

  game = engine.Engine(*art.shape, occlusion_in_layers=occlusion_in_layers)
  art = art.copy()  # frombuffer rows may be read-only
  for character in flat_schedule:
    game.update_group(group_name[character])
    mask = art == ord(character)
    if character in drapes:
      spec = drapes[character]
      game.add_prefilled_drape(character, mask, spec.pycolab_thing,
                               *spec.args, **spec.kwargs)
    if character in sprites:
      where = np.argwhere(mask)
      if len(where) > 1:
        raise ValueError('sprite character {} can appear in at most one place '
                         'in art.'.format(character))
      position = (int(where[0][0]), int(where[0][1])) if len(where) else (0, 0)
      spec = sprites[character]
      game.add_sprite(character, position, spec.pycolab_thing,
                      *spec.args, **spec.kwargs)
    art[mask] = beneath[mask]

  game.set_z_order(z_order)
  game.set_prefilled_backdrop(
      characters=''.join(chr(c) for c in np.unique(art)),
      prefill=art.view(np.uint8),
      backdrop_class=backdrop.pycolab_thing,
      *backdrop.args, **backdrop.kwargs)
  return game
