"""`Plot`: the per-game blackboard (reference: plot.py:27-385).

On the device the engine-facing part of the Plot (frame counter, summed
reward, discount, game-over: plot.py:69-104, 274-277) is a handful of
per-environment scalars in HBM.  The free-form dictionary part stays a host
dict; device programs keep their own blackboard entries (e.g. the scrolling
protocol's order/permits) in the per-environment state words.
"""


class Plot(dict):

  def __init__(self, engine=None):
    super(Plot, self).__init__()
    self._engine = engine
    self._update_group = None
    # global story state (plot.py:282-330): maintained by storytelling.Story
    self._prior_chapter = None
    self._this_chapter = None
    self._next_chapter = None
    self._device_next_at_set = None  # what the entities' word held when the host last assigned next_chapter

  @property
  def prior_chapter(self):
    """Key/index of the prior game in a `Story`, or None (plot.py:284-287)."""
    return self._prior_chapter

  @property
  def this_chapter(self):
    """Key/index of the current game in a `Story` (plot.py:289-292)."""
    return self._this_chapter

  @property
  def next_chapter(self):
    """Key/index of the next game in a `Story`, or None (plot.py:294-297).  "The last
    call before termination determines what happens" (plot.py:310-311): what a game
    entity assigned on the device (a `('next_chapter', key)` directive of a tabled
    entity) counts unless the host assigned the attribute AFTER it -- the setter notes
    what the entities' word held at that moment, and only a word that has changed since
    is a later assignment.  (An entity that re-assigns the very value its word already
    held after a host assignment cannot be told from no assignment; the host's value
    stands then.)"""
    v = self._device_next()
    if v is not None and v != self._device_next_at_set:
      from pycolab_amd import _native as N
      return None if v == N.CHAPTER_NONE else self._engine.chapter_key(v)  # (examples/ordeal.py names chapters by strings)
    return self._next_chapter

  def _device_next(self):
    """The entities' next_chapter word of a batch-1 game in play (synchronises), or None."""
    eng = self._engine
    if eng is not None and eng._native is not None and eng.batch == 1 and eng._assigns_next_chapter():
      from pycolab_amd import _native as N
      v = int(eng.entities_next_chapter()[0])
      return None if v == N.CHAPTER_UNSET else v
    return None

  @next_chapter.setter
  def next_chapter(self, next_chapter):
    """plot.py:299-324.  From the host (between `play()` calls); entities assign it on
    the device with a `('next_chapter', key)` directive (prefab_parts/tabled.py)."""
    self._next_chapter = next_chapter
    self._device_next_at_set = self._device_next()  # (later entity assignments change the word and win again)

  @property
  def frame(self):
    """Game iteration counter; an int for batch 1, else an int32 array [B]."""
    eng = self._engine
    if eng is None or eng._native is None:
      return -1
    frames = eng._read_scalars()['frame']
    return int(frames[0]) if eng.batch == 1 else frames

  @property
  def update_group(self):
    return self._update_group

  def log(self, message):
    del message  # device programs emit no strings (protocols/logging.py)

  # Step-time directives are issued by device programs, not from the host.
  def _host_directive(self, *unused_args, **unused_kwargs):
    raise NotImplementedError(
        'Plot directives are issued by device programs during the step kernel')

  add_reward = terminate_episode = change_z_order = _host_directive
