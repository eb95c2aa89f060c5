"""`Story`: a game made of other games (reference: pycolab/storytelling.py:36-475).

Same constructor, `its_showtime()` / `play()` contract and properties as the
reference: chapters are argumentless builders returning engines ready for
`its_showtime()`; a list/tuple plays them in order, a dict follows
`the_plot.next_chapter`; when a chapter terminates the next one is started at
once, its first observation and discount replace the old game's last ones and
the rewards add up (storytelling.py:391-470).  Everything here is host-side
orchestration of device engines; nothing of it is on the step path.

Batch 1 (the builders' engines have `batch == 1`) is the reference's behaviour
to the letter: one engine at a time, a fresh one from the builder whenever a
chapter starts.

Batch > 1 adds what the reference cannot express: every environment is in its
OWN chapter.  The story keeps one engine per chapter key, each over the whole
batch; an environment that enters a chapter is restarted there with a masked
reset (`Engine.reset(mask)`), only the environments that currently are in a
chapter read that chapter's results, and the per-environment observation is
assembled (on the device) over the union of the chapters' characters.  Engines
of chapters an environment is not in keep stepping it with nobody looking --
simple and correct, at the price of one launch per live chapter per step.
`the_plot.next_chapter` (plot.py:299-324) is written either by entities on the
device -- a `('next_chapter', key)` directive of a tabled entity, per environment,
as examples/ordeal.py:177-235 does from inside `update()` -- or from the host
(`story.the_plot.next_chapter = k` at batch 1, `story.set_next_chapter(k_or_array)`
at batch > 1).

Round 6: examples/ordeal.py itself.  The Plot entries its entities keep (`has_sword`,
`last_position`) and `the_plot.prior_chapter` are plot words on the device
(include/pcx.h PCX_PLOT_WORDS): what `new_plot.update(old_plot)` does for the
reference (storytelling.py:449-453) is, here, `plot_words()` of the engine an
environment leaves staged with `set_plot_words()` for the masked reset of the
engine it enters; rewards are float32 where a chapter's template says so; the
chapter codes its programs assign map to the Story's string keys through the
template's key table; chapters are built under the `engine.defaults(batch=...)`
the Story itself was built under.
"""

import collections

import numpy as np

from pycolab_amd import _native as _N
from pycolab_amd import cropping
from pycolab_amd import device as dev
from pycolab_amd import engine
from pycolab_amd import rendering
from pycolab_amd import things


class Story(object):
  """A programmable sequence of mutually compatible games."""

  def __init__(self, chapters, first_chapter=None, croppers=None, auto_reset=False):
    """As the reference's constructor (storytelling.py:105-170).  `auto_reset`
    (batch > 1 only): an environment whose story is over starts it again from
    the first chapter at the next `play()`, which counts as that step."""
    self._auto_reset = bool(auto_reset)
    self._auto_advance = not isinstance(chapters, collections.abc.Mapping)
    if self._auto_advance and first_chapter is None:
      first_chapter = 0
    self._chapters, self._croppers = _normalise(chapters, first_chapter, croppers)
    # the chapters are built later, under the engine defaults this Story was built under (`engine.defaults(batch=...)`
    # around an unchanged `make_game()`: examples/ordeal.py:82-110), and know their key from the start -- a device
    # program that asks the_plot.this_chapter (ordeal.py:216-262) has it compiled in
    self._defaults = engine.current_defaults()
    builders = dict(self._chapters)

    def build(key):
      with engine.defaults(**self._defaults):
        game = builders[key]()
      game.the_plot._this_chapter = key
      return game
    self._chapters = {key: (lambda key=key: build(key)) for key in builders}
    facts = _collect_facts(self._chapters, self._croppers)
    (self._chars_sprites, self._chars_drapes, self._chars_backdrops, (self._rows, self._cols), self._batch) = facts
    self._first_chapter = first_chapter
    self._showtime = False
    self._game_over = False
    self._dummies = {}
    if self._batch > 1:
      # every chapter keeps its own engine alive, so every chapter needs its own
      # cropper: one cropper object handed in for all chapters (which the
      # reference allows, storytelling.py:129-137) becomes one clone per chapter
      # -- a device cropper is bound to ONE engine and holds that engine's
      # per-environment window state
      seen = {}
      for key in sorted(self._croppers, key=repr):
        cr = self._croppers[key]
        if type(cr) is not cropping.ObservationCropper and id(cr) in seen:
          self._croppers[key] = cr._clone()
        seen[id(cr)] = key
    if self._batch == 1:
      self._current_game = self._chapters[first_chapter]()
      self._current_cropper = self._croppers[first_chapter]
      self._current_cropper.set_engine(self._current_game)
      self._stamp(self._current_game.the_plot, None, first_chapter)
    else:
      self._keys = sorted(self._chapters, key=repr)
      self._engines = {}
      self._chapter_of = np.full((self._batch,), self._keys.index(first_chapter), np.int32)  # -1: story over
      self._next_override = None
      self._entity_next_at_override = {}
      self._reward_float = False
      self._scalar_cache = {}
      # Layers with occlusion are `board == character` (rendering.py:177-179) and a cropper that never pads keeps them so:
      # while every chapter so far is like that the story assembles only the BOARD per chapter (one select per live chapter)
      # and derives all layers of the union in one compare (`_flush_layers`); the first chapter that is not ends that
      self._layers_from_board = True
      self._layers_stale = False
      self._union = sorted(self._chars_sprites | self._chars_drapes | self._chars_backdrops)
      self._current_game = self._engine_for(first_chapter)

  # ------------------------------------------------------------------ batch 1
  def _stamp(self, plot, prior, this):
    plot._prior_chapter, plot._this_chapter = prior, this
    if self._auto_advance:
      plot._next_chapter = this + 1 if (this + 1) in self._chapters else None

  def its_showtime(self):
    """storytelling.py:172-214."""
    if self._showtime:
      raise RuntimeError('its_showtime should not be called more than once.')
    self._showtime = True
    if self._batch > 1:
      return self._batched_start()
    observation, reward, discount = self._current_game.its_showtime()
    observation = self._current_cropper.crop(observation)
    if self._current_game.game_over:
      return self._start_next_game(observation, reward, discount)
    return observation, reward, discount

  def play(self, actions):
    """storytelling.py:216-283."""
    if not self._showtime:
      raise RuntimeError('play() cannot be called until the Story is placed in '
                         '"play mode" via the its_showtime() method.')
    if self._batch > 1:
      return self._batched_play(actions)
    if self._game_over:
      raise RuntimeError('play() was called after the last game managed by the '
                         'Story has terminated.')
    observation, reward, discount = self._current_game.play(actions)
    observation = self._current_cropper.crop(observation)
    if self._current_game.game_over:
      return self._start_next_game(observation, reward, discount)
    return observation, reward, discount

  def _start_next_game(self, observation, reward, discount):
    """storytelling.py:391-470: chain games until one survives its first frame."""
    while True:
      old_plot = self._current_game.the_plot
      nxt = old_plot.next_chapter
      if nxt is None:
        self._game_over = True
        return observation, reward, discount
      if nxt not in self._chapters:
        raise KeyError(
            'The game that just finished in the Story currently underway '
            '(identified by the key/index "{}") said that the next game in the '
            'story should be {}, but no game was supplied to the Story '
            'constructor under that key or index.'.format(old_plot.this_chapter, repr(nxt)))
      new_game = self._chapters[nxt]()
      new_game.the_plot.update(old_plot)  # values left in the Plot travel on (storytelling.py:449-450)
      self._stamp(new_game.the_plot, old_plot.this_chapter, nxt)
      self._current_game.close()
      self._current_game, self._current_cropper = new_game, self._croppers[nxt]
      self._current_cropper.set_engine(new_game)
      observation, more, discount = new_game.its_showtime()
      observation = self._current_cropper.crop(observation)
      if more is not None:
        reward = more if reward is None else reward + more
      if not new_game.game_over:
        return observation, reward, discount

  # ---------------------------------------------------------------- batch > 1
  def _engine_for(self, key):
    if key not in self._engines:
      game = self._chapters[key]()
      if game.batch != self._batch:
        raise ValueError('every chapter of a Story must have the same batch size')
      game._auto_reset = True  # environments that are not in this chapter run unobserved
      self._croppers[key].set_engine(game)
      self._engines[key] = game
      if not game.template.occlusion_in_layers or getattr(self._croppers[key], '_pad_char', None) is not None:
        self._flush_layers()
        self._layers_from_board = False
      if game.template.reward_is_float and not self._reward_float:  # (ordeal.py:123, 187-190 adds floats)
        self._reward_float = True
        if hasattr(self, '_reward'):
          self._reward = self._reward.astype(np.float64)
    return self._engines[key]

  def set_next_chapter(self, key):
    """Batch > 1: where environments go when their current chapter ends -- one
    key for all, or a sequence of keys per environment (None = the story ends
    for it).  Overrides the list order until changed again; `None` as the
    whole argument restores the default."""
    if key is None or not isinstance(key, (list, tuple, np.ndarray)):
      self._next_override = None if key is None else [key] * self._batch
    else:
      if len(key) != self._batch:
        raise ValueError('one next chapter per environment')
      self._next_override = list(key)
    # "The last call before termination determines what happens" (plot.py:310-311), as at batch 1 (plot.py here):
    # note what the entities' next_chapter words hold now -- only a word that changes afterwards is a later assignment
    self._entity_next_at_override = {}
    if self._next_override is not None:
      for k, eng in self._engines.items():
        if eng._native is not None and eng._assigns_next_chapter():
          self._entity_next_at_override[k] = eng.entities_next_chapter().copy()

  def _next_of(self, env, chapter_index, assigned=None):
    """Where environment `env` goes after chapter `chapter_index`: what the host said
    (`set_next_chapter`), else what the chapter's entities assigned to
    `the_plot.next_chapter` (plot.py:299-324; `assigned`: the engine's per-environment
    values), else the next chapter of a list -- and of the first two whichever spoke
    LAST (the rule of batch 1: plot.py's next_chapter getter)."""
    entity = assigned is not None and assigned[env] != _N.CHAPTER_UNSET
    if self._next_override is not None:
      before = self._entity_next_at_override.get(self._keys[chapter_index])
      later = entity and (before is None or before[env] != assigned[env])  # an entity spoke after the host did
      if not later:
        return self._next_override[env]
    if entity:  # (a chapter code: the key itself, or an index into the key table of the programs -- examples/ordeal.py)
      if assigned[env] == _N.CHAPTER_NONE:
        return None
      eng = getattr(self, '_engines', {}).get(self._keys[chapter_index])
      return int(assigned[env]) if eng is None else eng.chapter_key(assigned[env])
    if not self._auto_advance:
      return None
    nxt = self._keys[chapter_index] + 1
    return nxt if nxt in self._chapters else None

  def _batched_start(self):
    B = self._batch
    torch = dev.torch_module()
    L = len(self._union)
    self._planes = torch.zeros((B, 1 + L, self._rows, self._cols), dtype=torch.uint8, device='cuda:%d' % self._current_game._device_id)
    self._reward = np.zeros((B,), np.float64 if self._reward_float else np.int64)
    self._reward_set = np.zeros((B,), bool)
    self._discount = np.ones((B,), np.float32)
    first = self._keys.index(self._first_chapter)
    game = self._engine_for(self._first_chapter)
    obs, _, _ = game.its_showtime()
    self._started = {self._first_chapter}
    self._absorb(self._first_chapter, obs, np.ones((B,), bool))
    self._chain(self._finished(self._first_chapter, np.ones((B,), bool)))
    del first
    return self._batched_result()

  def _scalars(self, key):
    eng = self._engines[key]
    cached = self._scalar_cache.get(key)
    if cached is not None and cached[0] == eng._steps_launched:  # (the same launch: _absorb and _finished read it once)
      return cached[1]
    out = self._scalars_now(key)
    self._scalar_cache[key] = (eng._steps_launched, out)
    return out

  def _scalars_now(self, key):
    sc = self._engines[key]._read_scalars()  # one synchronisation: the story decides on the host
    rtype = np.float64 if self._engines[key].template.reward_is_float else np.int64
    return sc['reward'].astype(rtype), sc['reward_set'].astype(bool), sc['discount'], sc['done'].astype(bool)

  def _absorb(self, key, obs, members):
    """Takes chapter `key`'s results for the environments in `members`."""
    torch = dev.torch_module()
    reward, rset, discount, _ = self._scalars(key)
    self._reward[members] += np.where(rset[members], reward[members], 0)
    self._reward_set[members] |= rset[members]
    self._discount[members] = discount[members]
    cropped = self._croppers[key].crop(obs)
    if self._layers_from_board:
      if members.any():
        sel = torch.from_numpy(members).to(self._planes.device)[:, None, None]
        self._planes[:, 0] = torch.where(sel, _as_tensor(cropped.board, self._planes), self._planes[:, 0])
        self._layers_stale = True
      return
    idx = torch.from_numpy(np.flatnonzero(members)).to(self._planes.device)
    if idx.numel() == 0:
      return
    self._planes[idx, 0] = _as_tensor(cropped.board, self._planes)[idx]
    self._planes[idx, 1:] = 0
    for ch, layer in cropped.layers.items():
      self._planes[idx, 1 + self._union.index(ch)] = _as_tensor(layer, self._planes)[idx].to(torch.uint8)

  def _flush_layers(self):
    """The layers of the union's characters from the assembled boards (see `_layers_from_board`)."""
    if getattr(self, '_layers_stale', False):
      torch = dev.torch_module()
      if not hasattr(self, '_union_codes'):
        self._union_codes = torch.tensor([ord(c) for c in self._union], dtype=torch.uint8, device=self._planes.device)[None, :, None, None]
      self._planes[:, 1:].copy_(self._planes[:, :1] == self._union_codes)
      self._layers_stale = False

  def _finished(self, key, members):
    _, _, _, done = self._scalars(key)
    return {key: members & done}

  def _chain(self, finished):
    """Moves every environment whose chapter ended to its next one, restarting
    it there; repeats while first frames terminate (storytelling.py:391-470)."""
    while any(m.any() for m in finished.values()):
      starts = collections.defaultdict(lambda: np.zeros((self._batch,), bool))
      carry = None  # plot words that travel with the environments (storytelling.py:449-450: new_plot.update(old_plot))
      for key, mask in finished.items():
        ci = self._keys.index(key)
        eng = self._engines[key]
        assigned = eng.entities_next_chapter() if mask.any() and eng._assigns_next_chapter() else None
        if mask.any() and eng.template.n_plot_words:
          carry = _fresh_plot_words(self._batch) if carry is None else carry
          carry[:, mask] = eng.plot_words()[:, mask]
          keys = eng.template.chapter_keys or []
          carry[_N.PLOT_OD_PRIOR_CHAPTER, mask] = keys.index(key) if key in keys else -1  # new_plot.prior_chapter (:453)
        envs = np.flatnonzero(mask)
        if self._next_override is None and envs.size:
          # nobody on the host spoke (the common case): `_next_of` for all of them at once -- the entities' word through
          # the chapter's key table, else the next chapter of a list
          if assigned is None:
            codes = np.full(envs.shape, _N.CHAPTER_UNSET, np.int64)
          else:
            codes = assigned[envs].astype(np.int64)
          follow = None
          if self._auto_advance:
            follow = self._keys[ci] + 1 if (self._keys[ci] + 1) in self._chapters else None
          for code in np.unique(codes):
            these = envs[codes == code]
            nxt = follow if code == _N.CHAPTER_UNSET else None if code == _N.CHAPTER_NONE else eng.chapter_key(code)
            if nxt is None:
              self._chapter_of[these] = -1
            else:
              if nxt not in self._chapters:
                raise KeyError('no chapter {!r} was supplied to the Story constructor'.format(nxt))
              self._chapter_of[these] = self._keys.index(nxt)
              starts[nxt][these] = True
          continue
        for env in envs:
          nxt = self._next_of(env, ci, assigned)
          if nxt is None:
            self._chapter_of[env] = -1
          else:
            if nxt not in self._chapters:
              raise KeyError('no chapter {!r} was supplied to the Story constructor'.format(nxt))
            self._chapter_of[env] = self._keys.index(nxt)
            starts[nxt][env] = True
      finished = {}
      for key, mask in starts.items():
        game = self._engine_for(key)
        if key not in self._started and game.template.n_plot_words:
          game.its_showtime()  # (creates the device engine; the environments of `mask` start over below, with their plot words)
          self._started.add(key)
        if key in self._started:
          if game.template.n_plot_words:
            game.set_plot_words(_fresh_plot_words(self._batch) if carry is None else carry, mask)
          obs, _, _ = game.reset(mask)
        else:  # the engine's first reset covers every environment; only `mask` is looked at
          obs, _, _ = game.its_showtime()
          self._started.add(key)
        self._absorb(key, obs, mask)
        finished.update(self._finished(key, mask))

  def _batched_play(self, actions):
    live = self._chapter_of >= 0
    self._reward[:] = 0
    self._reward_set[:] = False
    self._discount[~live] = 0.0  # a finished story reports an empty step, like a finished engine (pcx.h)
    finished = {}
    restart = ~live if self._auto_reset else np.zeros_like(live)
    for ci in np.unique(self._chapter_of[live]):
      key = self._keys[ci]
      members = (self._chapter_of == ci) & live
      obs, _, _ = self._engines[key].play(actions)
      self._absorb(key, obs, members)
      finished[key] = self._finished(key, members)[key]
    if restart.any():  # a new story for these environments: frame 0 of the first chapter is their step
      first = self._first_chapter  # (after the engines have stepped: a restarted environment must not be stepped too)
      self._chapter_of[restart] = self._keys.index(first)
      if self._engines[first].template.n_plot_words:  # a new story has a new Plot
        self._engines[first].set_plot_words(_fresh_plot_words(self._batch), restart)
      obs, _, _ = self._engines[first].reset(restart)
      self._absorb(first, obs, restart)
      finished[first] = finished.get(first, np.zeros_like(live)) | self._finished(first, restart)[first]
    self._chain(finished)
    return self._batched_result()

  def _batched_result(self):
    self._flush_layers()
    layers = {ch: self._planes[:, 1 + k] for k, ch in enumerate(self._union)}
    obs = rendering.Observation(board=self._planes[:, 0], layers=layers)
    self._game_over = bool((self._chapter_of < 0).all())
    return obs, np.where(self._reward_set, self._reward, 0).astype(np.float32 if self._reward_float else np.int32), self._discount.copy()

  @property
  def reward_set(self):
    """Batch > 1: bool [B], False where the reference's reward would be None."""
    return self._reward_set.copy()

  @property
  def this_chapter(self):
    """The chapter key every environment is in (None where the story is over);
    batch 1: the current game's key."""
    if self._batch == 1:
      return self._current_game.the_plot.this_chapter
    return [None if c < 0 else self._keys[c] for c in self._chapter_of]

  # ---------------------------------------------------------------- properties
  @property
  def the_plot(self):
    return self._current_game.the_plot

  @property
  def rows(self):
    return self._rows

  @property
  def cols(self):
    return self._cols

  @property
  def batch(self):
    return self._batch

  @property
  def game_over(self):
    """bool for batch 1; for batch > 1 a bool array [B] (story over per environment)."""
    if self._batch == 1 or not self._showtime:
      return self._game_over
    return self._chapter_of < 0

  @property
  def z_order(self):
    """storytelling.py:308-322: unused characters first, then the current game's order."""
    current = self._current_game.z_order
    leftover = sorted((self._chars_sprites - set(current)) | (self._chars_drapes - set(current)))
    return leftover + current

  @property
  def backdrop(self):
    """storytelling.py:326-342."""
    return things.Backdrop(curtain=self._current_game.backdrop.curtain if self._current_game.backdrop else None,
                           palette=engine.Palette(self._chars_backdrops))

  @property
  def things(self):
    """storytelling.py:344-377: the current game's entities, plus stand-ins for
    the characters only other chapters use (`is_fictional` tells them apart).
    Batch > 1: the entities are the FIRST chapter engine's batched views (every
    environment has its own chapter: `this_chapter`, `engine_of(key).things`)."""
    out = dict(self._current_game.things)
    shape = (self._current_game.rows, self._current_game.cols)
    for c in self._chars_sprites:
      if c not in out:
        out[c] = self._dummies.setdefault(('s', c, shape), _DummySprite(things.Sprite.Position(*shape), c))
    for c in self._chars_drapes:
      if c not in out:
        out[c] = self._dummies.setdefault(('d', c, shape), _DummyDrape(np.zeros(shape, dtype=bool), c))
    return out

  @property
  def current_game(self):
    """storytelling.py:379-388.  Batch > 1: the first chapter's engine -- `the_plot`,
    `z_order` and `backdrop` describe that engine whichever chapters the
    environments are in; use `engine_of(key)` for another chapter's."""
    return self._current_game

  def engine_of(self, key):
    """Batch > 1: the engine that steps chapter `key` (created on first use)."""
    if self._batch == 1:
      raise RuntimeError('engine_of() is for batched stories; use current_game')
    if key not in self._chapters:
      raise KeyError(key)
    return self._engine_for(key)

  def close(self):
    for game in ([self._current_game] if self._batch == 1 else list(self._engines.values())):
      game.close()


def _fresh_plot_words(batch):
  """The plot words of a new Plot (include/pcx.h PCX_PLOT_OD_*): no sword, no last position, no prior chapter."""
  words = np.zeros((_N.PLOT_WORDS, batch), np.int32)
  words[_N.PLOT_OD_LAST_POSITION] = words[_N.PLOT_OD_PRIOR_CHAPTER] = -1
  return words


def is_fictional(thing):
  """True iff `thing` is one of the stand-ins `Story.things` returns
  (storytelling.py:473-483)."""
  return isinstance(thing, (_DummySprite, _DummyDrape))


class _DummySprite(things.Sprite):
  """An invisible Sprite standing for a character the current game does not use
  (storytelling.py:627-641)."""

  def __init__(self, corner, character):
    super(_DummySprite, self).__init__(corner=corner, position=self.Position(0, 0), character=character)
    self._visible = False

  def update(self, *args, **kwargs):
    raise RuntimeError('_DummySprite.update should never be called.')


class _DummyDrape(things.Drape):
  """An empty Drape standing for a character the current game does not use
  (storytelling.py:644-656)."""

  def update(self, *args, **kwargs):
    raise RuntimeError('_DummyDrape.update should never be called.')


def _as_tensor(x, like):
  torch = dev.torch_module()
  return x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x), device=like.device)


def _normalise(chapters, first_chapter, croppers):
  """Argument checks of storytelling.py:493-553."""
  if not chapters:
    raise ValueError('The chapters argument to the Story constructor must not be empty.')
  if isinstance(chapters, (list, tuple)):
    chapters = dict(enumerate(chapters))
    if isinstance(croppers, (list, tuple)):
      croppers = dict(enumerate(croppers))
  if not isinstance(chapters, collections.abc.Mapping):
    raise ValueError('The chapters argument to the Story constructor must be either a dict or a list.')
  if None in chapters:
    raise ValueError('None may not be a key in a Story chapters dict.')
  if first_chapter not in chapters:
    raise ValueError('The key "{}", specified as a Story\'s first_chapter, does not appear in '
                     'the chapters supplied to the Story constructor.'.format(first_chapter))
  if croppers is None:
    croppers = cropping.ObservationCropper()
  if isinstance(croppers, cropping.ObservationCropper):
    croppers = {k: croppers for k in chapters}
  if not isinstance(croppers, collections.abc.Mapping) or set(chapters) != set(croppers):
    raise ValueError('Since the croppers argument to the Story constructor was not None '
                     'or a single ObservationCropper, it must be a collection with the '
                     'same keys or indices as the chapters argument.')
  croppers = {k: cropping.ObservationCropper() if c is None else c for k, c in croppers.items()}
  return dict(chapters), croppers


def _collect_facts(chapters, croppers):
  """Compatibility checks of storytelling.py:556-624 (every builder is called
  once and its game started, then discarded)."""
  shapes, batches = set(), set()
  chars_sprites, chars_drapes, chars_backdrops = set(), set(), set()
  for key in sorted(chapters, key=repr):
    game = chapters[key]()
    cropper = croppers[key]
    cropper.set_engine(game)
    if game.backdrop is None and game._template is not None:  # Engine.from_template: the compiled game is all there is
      t = game._template
      kinds = {chr(sp['ch']): True for sp in t.sprites}
      kinds.update({chr(dr['ch']): False for dr in t.drapes})
      chars_backdrops.update(chr(c) for c in np.unique(t.backdrop))
    else:
      kinds = {ch: isinstance(thing, things.Sprite) for ch, thing in game.things.items()}
      chars_backdrops.update(game.backdrop.palette)
    observation, _, _ = game.its_showtime()
    board = cropper.crop(observation).board
    shapes.add(tuple(board.shape[-2:]))
    batches.add(game.batch)
    for ch, is_sprite in kinds.items():
      (chars_sprites if is_sprite else chars_drapes).add(ch)
    cropper.set_engine(None) if type(cropper) is cropping.ObservationCropper else cropper._release()
    game.close()
  if len(shapes) != 1:
    raise ValueError(
        'All pycolab games supplied to the Story constructor should have '
        'observations that are the same shape, either naturally or with the help '
        'of observation croppers. The games provided to the constructor have '
        'diverse shapes: {}.'.format(list(shapes)))
  if len(batches) != 1:
    raise ValueError('All games of a Story must have the same batch size; got {}.'.format(sorted(batches)))
  sd, sb, db = chars_sprites & chars_drapes, chars_sprites & chars_backdrops, chars_drapes & chars_backdrops
  if sd or sb or db:
    raise ValueError(
        'No two pycolab games supplied to the Story constructor should use the '
        'same character in two different ways: if a character is a Sprite in '
        'one game, it shouldn\'t be a Drape in another. Across the games '
        'supplied to this Story, these characters are both a Sprite and a '
        'Drape: [{}]; these are both a Sprite and in a Backdrop: [{}]; and '
        'these are both a Drape and in a Backdrop: [{}].'.format(*[''.join(sorted(s)) for s in (sd, sb, db)]))
  return chars_sprites, chars_drapes, chars_backdrops, shapes.pop(), batches.pop()
