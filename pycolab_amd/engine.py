"""The batched game engine facade.

Public surface of the reference's `pycolab/engine.py:38-986` (`Engine`,
`Palette`): the same builder methods, `its_showtime()`, `play()`, and
properties.  What differs is where the step runs: `its_showtime()` compiles
the built game into a plain-data template (`pycolab_amd.compiler`), creates a
HIP engine for `batch` environments through the C ABI (`include/pcx.h`), and
every `play()` is one launch of the fused step kernel over the whole batch.

Batch semantics.  `configure(batch=B)` selects how many independent copies of
the game are stepped together (default 1).  With `batch == 1` the return
values have exactly the reference's types and shapes.  With `batch > 1`
`Observation.board` is [B, rows, cols], each layer is [B, rows, cols], reward
is an int32 array [B] (0 where the reference would return `None`; see
`Engine.reward_set`), discount is a float32 array [B]; all are device tensors
when PyTorch-ROCm is available, NumPy arrays otherwise.
"""

import collections
import contextlib
import ctypes

import numpy as np

from pycolab_amd import _native as N
from pycolab_amd import compiler
from pycolab_amd import device as dev
from pycolab_amd import plot
from pycolab_amd import rendering
from pycolab_amd import things


_DEFAULTS = dict(batch=1, device=0)


@contextlib.contextmanager
def defaults(batch=None, device=None):
  """Engines built inside the block start with this batch size / GPU -- for game files that build their engines where
  the caller cannot reach them to `configure()`: `with engine.defaults(batch=4096): story = ordeal.make_game()` (the
  chapters of examples/ordeal.py:82-110 are built inside its Story).  A `Story` remembers the defaults it was built
  under and builds its chapters under them later."""
  saved = dict(_DEFAULTS)
  if batch is not None:
    if int(batch) < 1:
      raise ValueError('batch must be >= 1')
    _DEFAULTS['batch'] = int(batch)
  if device is not None:
    _DEFAULTS['device'] = int(device)
  try:
    yield
  finally:
    _DEFAULTS.clear()
    _DEFAULTS.update(saved)


def current_defaults():
  return dict(_DEFAULTS)


class Engine(object):
  """One game template stepped as `batch` independent environments on a GPU."""

  def __init__(self, rows, cols, occlusion_in_layers=True):
    self._rows = rows
    self._cols = cols
    self._occlusion_in_layers = occlusion_in_layers
    self._the_plot = plot.Plot(self)
    self._showtime = False
    self._backdrop = None
    self._sprites_and_drapes = collections.OrderedDict()  # z-order, back to front
    self._update_groups = collections.defaultdict(list)
    self._current_update_group = ''
    # batched runtime
    self._batch = _DEFAULTS['batch']
    self._device_id = _DEFAULTS['device']
    self._auto_reset = False
    self._seed = 0
    self._env_offset = 0
    self._pitch = rows * cols
    self._template = None
    self._native = None
    self._keepalive = None
    self._bufs = None
    self._packed = None
    self._actions = None
    self._steps_launched = 0  # reset / step launches so far (fused epilogues know when they are fresh)
    self._croppers = []
    self._fuse_request = None  # cropping.fuse_croppers() before its_showtime(): (croppers, only_crops)
    self._only_crops = False   # fused croppers only: the full-board planes are not written
    self._epilogue_only = False  # a fused post-processor with skip_board: the kernel writes its array only
    self._epilogue = None      # (converter, float tensor the step kernel writes): rendering.fuse_into

  def _install_epilogue(self, converter, out, only=False):
    """rendering.ObservationToFeatureArray.fuse_into(): the step kernel writes
    `out` from now on, so the engine holds it (and its converter) until the
    epilogue is cleared or the engine closed."""
    if self._epilogue is not None and self._epilogue[0] is not converter:
      self._epilogue[0]._epilogue_gone()  # the kernel feeds one array: the earlier converter is on its own again
    self._epilogue = (converter, out)
    self._epilogue_only = bool(only)

  def _clear_epilogue(self, converter=None):
    if self._epilogue is None or (converter is not None and self._epilogue[0] is not converter):
      return
    if self._native is not None:
      N.check(N.lib().pcx_engine_set_epilogue(self._native, None))
    self._epilogue[0]._epilogue_gone()
    self._epilogue = None
    self._epilogue_only = False

  def _register_cropper(self, cropper):
    if cropper not in self._croppers:
      self._croppers.append(cropper)
    if self._native is not None:
      cropper._create_native()

  # ---------------------------------------------------------------- builder API
  def _unregister_cropper(self, cropper):
    if cropper in self._croppers:
      self._croppers.remove(cropper)

  # The builder keeps the reference's surface (names, argument order, error texts: SURVEY 8b) around three steps of
  # its own: _claim() = "may this be called now, and are these characters free", _plane() = a private, exactly-typed
  # copy of a prefill, _adopt() = a new Sprite or Drape enters the z-order (in front) and the current update group.
  def _claim(self, method_name, characters, single=False):
    self._forbid_after_showtime(method_name)
    self._require_good_characters(characters, mandatory_len=1 if single else None)
    self._require_unclaimed(characters)

  def _plane(self, prefill, dtype):
    """A rows x cols array of `dtype` holding `prefill`, which must have that very dtype ('equiv' casting, as the
    reference copies it: engine.py:325-327, :408-410) and broadcast to the board."""
    plane = np.empty((self._rows, self._cols), dtype=dtype)
    np.copyto(plane, prefill, casting='equiv')
    return plane

  def _adopt(self, character, thing):
    self._sprites_and_drapes[character] = thing
    self._update_groups[self._current_update_group].append(thing)
    return thing

  def set_backdrop(self, characters, backdrop_class, *args, **kwargs):
    """engine.py:248-279: a backdrop whose curtain starts as all zeros."""
    blank = np.zeros((self._rows, self._cols), dtype=np.uint8)
    return self.set_prefilled_backdrop(characters, blank, backdrop_class, *args, **kwargs)

  def set_prefilled_backdrop(self, characters, prefill, backdrop_class,
                             *args, **kwargs):
    """engine.py:281-337."""
    self._claim('set_prefilled_backdrop', characters)
    if self._backdrop:
      raise RuntimeError('A backdrop of type {} has already been supplied to '
                         'this Engine.'.format(type(self._backdrop)))
    if not issubclass(backdrop_class, things.Backdrop):
      raise TypeError('backdrop_class arguments to Engine.set_backdrop must '
                      'either be a Backdrop class or one of its subclasses.')
    self._backdrop = backdrop_class(self._plane(prefill, np.uint8), Palette(characters), *args, **kwargs)
    return self._backdrop

  def add_drape(self, character, drape_class, *args, **kwargs):
    """engine.py:339-369: a drape whose curtain starts empty."""
    nothing = np.zeros((self._rows, self._cols), dtype=np.bool_)
    return self.add_prefilled_drape(character, nothing, drape_class, *args, **kwargs)

  def add_prefilled_drape(self, character, prefill, drape_class,
                          *args, **kwargs):
    """engine.py:371-421."""
    self._claim('add_prefilled_drape', character, single=True)
    if not issubclass(drape_class, things.Drape):
      raise TypeError('drape_class arguments to Engine.add_drape must be a '
                      'subclass of Drape')
    return self._adopt(character, drape_class(self._plane(prefill, np.bool_), character, *args, **kwargs))

  def add_sprite(self, character, position, sprite_class, *args, **kwargs):
    """engine.py:423-470."""
    self._claim('add_sprite', character, single=True)
    if not issubclass(sprite_class, things.Sprite):
      raise TypeError('sprite_class arguments to Engine.add_sprite must be a '
                      'subclass of Sprite')
    row, col = position[0], position[1]
    if not (0 <= row < self._rows and 0 <= col < self._cols):
      raise ValueError('Position {} does not fall inside a {}x{} game board.'
                       ''.format(position, self._rows, self._cols))
    board_corner = things.Sprite.Position(self._rows, self._cols)  # (what the reference hands a Sprite as `corner`)
    return self._adopt(character, sprite_class(board_corner, things.Sprite.Position(*position), character, *args, **kwargs))

  def update_group(self, group_name):
    """engine.py:472-489: what is added from now on updates in this group."""
    self._forbid_after_showtime('update_group')
    self._current_update_group = group_name

  def set_z_order(self, z_order):
    """engine.py:491-518: `z_order` back to front, every Sprite and Drape exactly once."""
    self._forbid_after_showtime('set_z_order')
    current = self._sprites_and_drapes
    if len(z_order) != len(current) or any(c not in current for c in z_order) or len(set(z_order)) != len(current):
      raise ValueError('The z_order argument {} to Engine.set_z_order is not a '
                       'proper permutation of the characters corresponding to '
                       'Sprites and Drapes in this game, which are {}.'.format(
                           repr(z_order), current.keys()))
    reordered = collections.OrderedDict()
    for character in z_order:
      reordered[character] = current[character]
    self._sprites_and_drapes = reordered

  def _frozen_update_groups(self):
    """Update groups sorted by name (engine.py:557-558)."""
    return [(k, self._update_groups[k]) for k in sorted(self._update_groups)]

  # ------------------------------------------------------------ batched runtime
  def configure(self, batch=None, device=None, auto_reset=None, seed=None, env_offset=None):
    """Choose how many environments to step, on which GPU, and what happens
    to environments whose episode has ended (`auto_reset=True`: the next
    step rebuilds them from the template and runs frame 0; counted as one
    env-step).  Must be called before `its_showtime()`."""
    self._forbid_after_showtime('configure')
    if batch is not None:
      if int(batch) < 1:
        raise ValueError('batch must be >= 1')
      self._batch = int(batch)
    if device is not None:
      self._device_id = int(device)
    if auto_reset is not None:
      self._auto_reset = bool(auto_reset)
    if seed is not None:        # games that draw random numbers (marauders)
      self._seed = int(seed)
    if env_offset is not None:  # global index of this engine's first environment
      self._env_offset = int(env_offset)
    return self

  @classmethod
  def from_template(cls, template, batch=1, device=0, auto_reset=False, seed=0, env_offset=0):
    """An engine for a pre-compiled `GameTemplate` (no Python entity objects)."""
    eng = cls(template.rows, template.cols, template.occlusion_in_layers)
    eng._template = template
    eng.configure(batch=batch, device=device, auto_reset=auto_reset, seed=seed, env_offset=env_offset)
    return eng

  @property
  def batch(self):
    return self._batch

  @property
  def template(self):
    if self._template is None:
      self._template = compiler.GameTemplate.from_engine(self)
    return self._template

  def its_showtime(self):
    """engine.py:520-581: start the episode(s); returns the frame-0 triple."""
    self._forbid_after_showtime('its_showtime')
    template = self.template  # compile before flipping any state
    lib = N.lib()
    # param[0..3]: RNG seed and global environment offset (64 bits each)
    template.param[0], template.param[1] = self._seed & 0xFFFFFFFF, (self._seed >> 32) & 0xFFFFFFFF
    template.param[2], template.param[3] = self._env_offset & 0xFFFFFFFF, (self._env_offset >> 32) & 0xFFFFFFFF
    ct, keep = template.to_ctypes()
    handle = ctypes.c_void_p()
    N.check(lib.pcx_engine_create(ctypes.byref(ct), self._batch,
                                  self._device_id, ctypes.byref(handle)))
    self._native, self._keepalive = handle, keep
    B, L, R, C = self._batch, len(template.chars), self._rows, self._cols
    self._pitch = int(lib.pcx_engine_plane_pitch(self._native))
    mk = lambda shape, dt: dev.DeviceBuffer(shape, dt, self._device_id)
    self._bufs = dict(planes=mk((B, 1 + L, self._pitch), np.uint8), frame=mk((B,), np.int32),
                      error=mk((B,), np.uint8))
    # what play() returns besides the observation, 10 bytes per environment, in
    # ONE allocation [reward i32 B | discount f32 B | reward_set u8 B | done u8 B]:
    # a multi-GPU consumer gathers it with a single collective and no packing
    # pass (pycolab_amd.distributed.ScalarGather)
    self._packed = None
    rdt = np.float32 if template.reward_is_float else np.int32  # pcx_template::reward_is_float: the lane holds float32 bits
    if dev.torch_module() is not None:
      self._packed = mk((10 * B,), np.uint8)
      cut = self._packed.tensor
      view = lambda lo, hi, dt: dev.DeviceBuffer.view_of(cut[lo:hi], dt, self._device_id)
      self._bufs.update(reward=view(0, 4 * B, rdt), discount=view(4 * B, 8 * B, np.float32),
                        reward_set=view(8 * B, 9 * B, np.uint8), done=view(9 * B, 10 * B, np.uint8))
    else:
      self._bufs.update(reward=mk((B,), rdt), discount=mk((B,), np.float32),
                        reward_set=mk((B,), np.uint8), done=mk((B,), np.uint8))
    self._actions = mk((B,), np.int32)
    ext = N.Buffers(batch=B, rows=R, cols=C, n_chars=L,
                    **{k: v.ptr for k, v in self._bufs.items()})
    N.check(lib.pcx_engine_bind_buffers(self._native, ctypes.byref(ext)))
    self._showtime = True
    self._current_update_group = None
    for cropper in self._croppers:  # croppers attached with set_engine() before showtime
      cropper._create_native()
    if self._fuse_request is not None:  # fused croppers see frame 0 like every other frame
      from pycolab_amd import cropping
      request, self._fuse_request = self._fuse_request, None
      cropping.fuse_croppers(self, *request)
    if template.n_plot_words and self._batch == 1:
      self._plot_to_device()  # what a Story copied into this game's Plot (storytelling.py:449-450) is what its programs start from
    N.check(lib.pcx_engine_reset(self._native, None, dev.current_stream(self._device_id)))
    self._steps_launched += 1
    return self._result()

  # ---- plot words: the Plot entries device programs use (include/pcx.h PCX_PLOT_WORDS; examples/ordeal.py) ----------
  def plot_words(self):
    """int32 [PLOT_WORDS, batch]: the plot words as the last step left them (`pcx_engine_plot_words`).  Synchronises."""
    self._b  # (raises after close())
    out = np.empty((N.PLOT_WORDS, self._batch), np.int32)
    N.check(N.lib().pcx_engine_plot_words(self._native, out.ctypes.data))
    return out

  def set_plot_words(self, words, env_mask=None):
    """The plot words the environments selected by `env_mask` (host bool [batch]; None: all) start their NEXT episode
    with (`pcx_engine_set_plot_words`) -- a Story's `new_plot.update(old_plot)` for device programs."""
    self._b
    words = np.ascontiguousarray(words, np.int32).reshape(N.PLOT_WORDS, self._batch)
    mask = None if env_mask is None else np.ascontiguousarray(env_mask, np.uint8)
    N.check(N.lib().pcx_engine_set_plot_words(self._native, words.ctypes.data, None if mask is None else mask.ctypes.data))

  def _plot_to_device(self):
    """Batch 1, examples/ordeal.py: the Plot dict's 'has_sword' / 'last_position' and the_plot.prior_chapter -> plot words."""
    plot_, keys = self._the_plot, self._template.chapter_keys or []
    words = np.zeros((N.PLOT_WORDS, 1), np.int32)
    words[N.PLOT_OD_HAS_SWORD] = int(bool(plot_.get('has_sword')))
    lp = plot_.get('last_position')
    words[N.PLOT_OD_LAST_POSITION] = -1 if lp is None else (int(lp[0]) & 0xFFFF) | (int(lp[1]) << 16)
    words[N.PLOT_OD_PRIOR_CHAPTER] = keys.index(plot_.prior_chapter) if plot_.prior_chapter in keys else -1
    self.set_plot_words(words)

  def _plot_from_device(self):
    """... and back after every step, so that `the_plot['has_sword']` reads as in the reference and travels with the dict."""
    words = self.plot_words()[:, 0]
    if words[N.PLOT_OD_HAS_SWORD]:
      dict.__setitem__(self._the_plot, 'has_sword', True)
    else:  # (the entry exists only once the sword has been picked up: ordeal.py:123; gone again after a reset in place)
      dict.pop(self._the_plot, 'has_sword', None)
    lp = int(words[N.PLOT_OD_LAST_POSITION])
    if lp != -1:
      r, c = lp & 0xFFFF, (lp >> 16) & 0xFFFF
      dict.__setitem__(self._the_plot, 'last_position',
                       things.Sprite.Position(r - 0x10000 if r >= 0x8000 else r, c - 0x10000 if c >= 0x8000 else c))
    else:
      dict.pop(self._the_plot, 'last_position', None)

  def play(self, actions):
    """engine.py:583-639: one step of every environment.

    `actions`: for batch 1 a scalar (or `None`); for batch > 1 `None`, a
    scalar (broadcast), an int array [B], or an int32 device tensor [B].
    """
    if not self._showtime:
      raise RuntimeError('play() cannot be called until the Engine is placed '
                         'in "play mode" via the its_showtime() method.')
    if self._batch == 1 and not self._auto_reset and self.game_over:
      raise RuntimeError('play() was called after the episode handled by this '
                         'Engine has terminated.')
    self.step(actions)
    if self._batch > 1:
      # no host synchronisation per step: ask (asynchronously) whether any
      # environment raised, and act on what an earlier step's poll found
      seen = N.c_i32(0)
      N.check(N.lib().pcx_engine_error_poll(self._native, dev.current_stream(self._device_id), ctypes.byref(seen)))
      if seen.value:
        self.check_errors()
    return self._result()

  def reset(self, env_mask=None):
    """Restart environments in place (a new episode = a new Engine in the
    reference): all of them, or those where `env_mask` (uint8/bool [batch],
    host array or device tensor) is nonzero."""
    self._b  # (raises after close())
    ptr, keep = None, None
    if env_mask is not None:
      torch = dev.torch_module()
      if torch is not None and isinstance(env_mask, torch.Tensor):
        keep = env_mask.to(dtype=torch.uint8).contiguous()
        ptr = keep.data_ptr()
      else:
        keep = dev.DeviceBuffer((self._batch,), np.uint8, self._device_id)
        keep.upload(np.asarray(env_mask, np.uint8))
        ptr = keep.ptr
    N.check(N.lib().pcx_engine_reset(self._native, ptr, dev.current_stream(self._device_id)))
    self._steps_launched += 1
    dev.synchronize(self._device_id)
    return self._result()

  def step(self, actions):
    """`play()` without materialising return values (no host sync)."""
    self._b  # (raises after close())
    ptr = self._stage_actions(actions)
    N.check(N.lib().pcx_engine_step(self._native, ptr, int(self._auto_reset),
                                    dev.current_stream(self._device_id)))
    self._steps_launched += 1

  def step_n(self, action_tape):
    """`len(action_tape)` consecutive steps from a device int32 tensor (or host
    array) of shape [T, batch]; only the last step's observation survives."""
    self._b  # (raises after close())
    torch = dev.torch_module()
    if torch is not None and isinstance(action_tape, torch.Tensor):
      if action_tape.dtype != torch.int32 or not action_tape.is_cuda or not action_tape.is_contiguous():
        raise ValueError('device action tape must be a contiguous int32 CUDA tensor [T, batch]')
      tape, ptr = action_tape, action_tape.data_ptr()
    else:
      host = np.ascontiguousarray(action_tape, np.int32)
      tape = dev.DeviceBuffer(host.shape, np.int32, self._device_id)
      tape.upload(host)
      ptr = tape.ptr
    steps, batch = tape.shape
    if batch != self._batch:
      raise ValueError('action tape must have shape [T, batch]')
    N.check(N.lib().pcx_engine_step_n(self._native, ptr, int(steps), int(self._auto_reset),
                                      dev.current_stream(self._device_id)))
    self._steps_launched += int(steps)
    dev.synchronize(self._device_id)  # the staging buffer may be freed after this call

  def step_hashed(self, seed, t0, steps, env_offset=None):
    """`steps` steps with on-device actions `hash(seed, env, t) % n_actions`,
    env = `env_offset` (default: the engine's configured offset) + local index."""
    self._b  # (raises after close())
    if env_offset is None:
      env_offset = self._env_offset
    N.check(N.lib().pcx_engine_step_hashed(
        self._native, seed, env_offset, t0, steps, int(self._auto_reset),
        dev.current_stream(self._device_id)))
    self._steps_launched += int(steps)

  def _stage_actions(self, actions):
    torch = dev.torch_module()
    if torch is not None and isinstance(actions, torch.Tensor):
      if (actions.dtype != torch.int32 or not actions.is_cuda or
          actions.numel() != self._batch or not actions.is_contiguous()):
        raise ValueError('device actions must be a contiguous int32 CUDA tensor [batch]')
      return actions.data_ptr()
    if actions is None:
      host = np.full((self._batch,), N.ACTION_NONE, np.int32)
    elif np.isscalar(actions) or np.ndim(actions) == 0:
      host = np.full((self._batch,), int(actions), np.int32)
    else:
      host = np.asarray([N.ACTION_NONE if a is None else int(a) for a in actions]
                        if isinstance(actions, (list, tuple)) else actions,
                        dtype=np.int32).reshape(self._batch)
    self._actions.upload(host)
    return self._actions.ptr

  def _read_scalars(self):
    dev.synchronize(self._device_id)
    return {k: self._b[k].numpy() for k in
            ('reward', 'reward_set', 'discount', 'done', 'frame', 'error')}

  def entities_next_chapter(self):
    """int32 [batch]: what every environment's entities last assigned to
    `the_plot.next_chapter` in its current episode (a `('next_chapter', key)`
    directive: plot.py:299-324) -- the chapter's index, `_native.CHAPTER_NONE` for
    None, `_native.CHAPTER_UNSET` where no entity has.  Synchronises."""
    self._b  # (raises after close())
    out = np.empty((self._batch,), np.int32)
    N.check(N.lib().pcx_engine_next_chapter(self._native, out.ctypes.data))
    return out

  def _assigns_next_chapter(self):
    t = self.template
    return any(d[1] == N.DIR_NEXT_CHAPTER for d in t.directives) or any(
        s['program'] in (N.PROG_OD_PLAYER, N.PROG_OD_DRAGONDUCK) for s in t.sprites)  # examples/ordeal.py:177-246

  def chapter_key(self, code):
    """The Story key behind a chapter code the entities assigned (`entities_next_chapter`): the code itself (an integer
    key or list index) unless the template's programs name chapters by a key table (examples/ordeal.py: strings)."""
    keys = self.template.chapter_keys
    return int(code) if keys is None else keys[int(code)]

  def export_state(self, with_observation=False):
    """A checkpoint of every environment's episode as a NumPy uint8 array
    (`pcx_engine_export_state`): state words (entity state, Plot scalars, RNG
    draw counters) and what the last `play()` returned; `with_observation=True`
    adds the observation planes.  Synchronises."""
    self._b  # (raises after close())
    n = N.c_u64(0)
    N.check(N.lib().pcx_engine_state_size(self._native, int(bool(with_observation)), ctypes.byref(n)))
    blob = np.empty((n.value,), np.uint8)
    N.check(N.lib().pcx_engine_export_state(self._native, blob.ctypes.data, n.value, int(bool(with_observation))))
    # the croppers attached to this engine carry state of their own -- every environment's window corner, the margin
    # hysteresis of cropping.py:393-426 lives in it -- and, with the observation, their cropped planes: appended in
    # attachment order behind a trailer [b'PCXT', count, sizes...] (a resumed ScrollingCropper must not re-centre)
    parts = []
    for cropper in self._croppers:
      if getattr(cropper, '_native', None) is None:
        cropper._create_native()
      if getattr(cropper, '_native', None) is None:
        # the pass-through base ObservationCropper (what a Story installs by default) has no device object and no
        # state: a zero-length part keeps the attachment order of the trailer intact
        parts.append(np.empty((0,), np.uint8))
        continue
      m = N.c_u64(0)
      N.check(N.lib().pcx_cropper_state_size(cropper._native, int(bool(with_observation)), ctypes.byref(m)))
      part = np.empty((m.value,), np.uint8)
      N.check(N.lib().pcx_cropper_export_state(cropper._native, part.ctypes.data, m.value, int(bool(with_observation))))
      parts.append(part)
    trailer = np.array([0x54584350, len(parts)] + [p.size for p in parts], np.uint64).view(np.uint8)
    return np.concatenate([blob] + parts + [trailer, np.array([trailer.size], np.uint64).view(np.uint8)])

  def import_state(self, blob):
    """Restores a checkpoint made by an engine of the same template and batch
    (this one must be in play: `its_showtime()` has run).  The steps that follow
    are exactly the steps the exporting engine would have taken."""
    self._b  # (raises after close())
    blob = np.ascontiguousarray(blob, np.uint8)
    parts = []
    if blob.size >= 24:  # the croppers' trailer (export_state): [magic, count, sizes...] and its own length at the very end
      tsize = int(blob[-8:].view(np.uint64)[0])
      if 16 <= tsize <= blob.size - 8 and tsize % 8 == 0:
        words = blob[-8 - tsize:-8].view(np.uint64)
        if int(words[0]) == 0x54584350 and words.size == 2 + int(words[1]):
          sizes = [int(x) for x in words[2:]]
          end = blob.size - 8 - tsize
          for size in reversed(sizes):
            parts.insert(0, blob[end - size:end])
            end -= size
          blob = blob[:end]
    croppers = list(self._croppers)
    if len(parts) != len(croppers):
      raise ValueError('the checkpoint holds the state of {} cropper(s), this engine has {} attached: attach the same '
                       'croppers, in the same order, before import_state()'.format(len(parts), len(croppers)))
    blob = np.ascontiguousarray(blob)
    N.check(N.lib().pcx_engine_import_state(self._native, blob.ctypes.data, blob.nbytes))
    for cropper, part in zip(croppers, parts):
      if getattr(cropper, '_native', None) is None:
        cropper._create_native()
      if getattr(cropper, '_native', None) is None:  # (a pass-through cropper: nothing to restore)
        if part.size:
          raise ValueError('the checkpoint holds window state for a cropper that has none here: attach the same croppers, in the same order')
        continue
      part = np.ascontiguousarray(part)
      N.check(N.lib().pcx_cropper_import_state(cropper._native, part.ctypes.data, part.nbytes))
    dev.synchronize(self._device_id)
    # a feature stack fused into a cropper's window (ObservationToFeatureArray.fuse_into(engine, source=cropper)) was
    # written by the exporting engine's step kernel: recompute it from the window as restored (or recut), so that the
    # converter does not hand out the pre-import tensor as this step's
    with_observation = bool(blob[48:52].view(np.int32)[0]) if blob.size >= 64 else False  # StateHeader.with_observation
    for cropper in croppers:
      feats = getattr(cropper, '_features', None)
      if feats is not None:
        feats[0]._window_after_import(cropper, feats[1], with_observation)
    # a fused post-processor's array was written by the exporting engine's step kernel, not by this one: refill it
    # from the restored observation where there is one; otherwise it counts as not written (the converter falls back
    # to its own kernel, which raises for an engine that writes no planes) until the next step
    if self._epilogue is not None:
      converter, out = self._epilogue
      converter._after_import(self, out, with_observation and not self._epilogue_only)

  def tuner_done(self):
    """False while the engine's first step launches still take turns measuring launch shapes (include/pcx.h
    pcx_engine_tuner_done); a benchmark steps until this is True before it times anything."""
    return bool(N.lib().pcx_engine_tuner_done(self._native))

  def check_errors(self):
    """Synchronises and raises if a device program hit a condition the
    reference raises for (the bits are sticky within an episode).  `play()`
    with batch 1 checks every step; with batch > 1 it polls asynchronously and
    raises one or two steps late; `step()` never checks."""
    self._b  # (raises after close())
    err = np.zeros((self._batch,), np.uint8)
    # the live error array ORed with what the asynchronous polls saw: an
    # environment that raised and was auto-reset since is still reported
    N.check(N.lib().pcx_engine_errors_seen(self._native, err.ctypes.data, 1))
    if err.any():
      bad = int(np.flatnonzero(err)[0])
      code = int(err[bad])
      kinds = [name for bit, name in ((1, 'IndexError'), (2, 'scrolling.Error')) if code & bit]
      raise RuntimeError('environment {} raised {} on the device ({} of {} environments affected)'.format(
          bad, kinds, int(np.count_nonzero(err)), err.size))

  def planes_view(self, host=False):
    """Observation planes as [B, 1+n_chars, rows, cols]: a zero-copy strided
    view of the device tensor (plane pitch may exceed rows*cols), or a NumPy
    copy with `host=True` / without PyTorch."""
    planes = self._b['planes']
    B, P, R, C = self._batch, 1 + len(self._template.chars), self._rows, self._cols
    if planes.tensor is not None and not host:
      return planes.tensor.as_strided((B, P, R, C), (P * self._pitch, self._pitch, C, 1))
    return np.ascontiguousarray(planes.numpy()[:, :, :R * C]).reshape(B, P, R, C)

  def _result(self):
    L = self._template.chars
    if self._only_crops or self._epilogue_only:
      # cropping.fuse_croppers(..., only_crops=True) / fuse_into(..., skip_board=True):
      # the step kernel no longer writes the full-board planes, so there is no
      # observation to hand out -- the fused croppers' crop() (the fused
      # converter's call) returns what the step wrote
      sc = None if self._batch > 1 else self._read_scalars()
      stale = self._tag(rendering.Observation(board=None, layers={}))
      stale._planes_stale = True  # (a post-processor or cropper that would read the engine's planes raises instead)
      if self._batch == 1:
        self.check_errors()
        if self._template.n_plot_words:
          self._plot_from_device()
        return stale, (self._reward_type(sc['reward'][0]) if sc['reward_set'][0] else None), float(sc['discount'][0])
      pick = lambda k: (self._b[k].tensor if self._b[k].tensor is not None else self._b[k].numpy())
      return stale, pick('reward'), pick('discount')
    if self._batch == 1:
      sc = self._read_scalars()
      self.check_errors()
      p = self.planes_view(host=True)[0]
      layers = {chr(c): p[1 + k].astype(np.bool_) for k, c in enumerate(L)}
      reward = self._reward_type(sc['reward'][0]) if sc['reward_set'][0] else None
      if self._template.n_plot_words:
        self._plot_from_device()
      return self._tag(rendering.Observation(board=p[0], layers=layers)), reward, float(sc['discount'][0])
    arr = self.planes_view()
    layers = {chr(c): arr[:, 1 + k] for k, c in enumerate(L)}
    pick = lambda k: (self._b[k].tensor if self._b[k].tensor is not None
                      else self._b[k].numpy())
    return (self._tag(rendering.Observation(board=arr[:, 0], layers=layers)),
            pick('reward'), pick('discount'))

  def _reward_type(self, value):
    return float(value) if self._template.reward_is_float else int(value)

  def _tag(self, observation):
    observation._source = self
    return observation

  def _planes_view(self):
    view = N.PlanesView()
    N.check(N.lib().pcx_engine_planes_view(self._native, ctypes.byref(view)))
    return view, self._device_id

  # ---------------------------------------------------------------- properties
  @property
  def planes(self):
    """The raw observation planes buffer [B, 1+n_chars, pitch] (see planes_view)."""
    return self._b['planes']

  @property
  def buffers(self):
    return self._b

  @property
  def scalars_packed(self):
    """uint8 device tensor [10 * batch]: reward (int32) | discount (float32) |
    reward_set | done, the storage behind `buffers[...]` (None without PyTorch)."""
    return None if self._packed is None else self._packed.tensor

  @property
  def reward_set(self):
    """uint8 [B]: 0 where the reference's reward would be `None`."""
    b = self._b['reward_set']
    return b.tensor if b.tensor is not None else b.numpy()

  @property
  def the_plot(self):
    return self._the_plot

  @property
  def rows(self):
    return self._rows

  @property
  def cols(self):
    return self._cols

  @property
  def game_over(self):
    """bool for batch 1; uint8 array [B] otherwise (engine.py:660-662)."""
    if self._native is None:
      return False
    dev.synchronize(self._device_id)
    done = self._b['done'].numpy()  # this one array only
    return bool(done[0]) if self._batch == 1 else done

  @property
  def z_order(self):
    if self._sprites_and_drapes or self._template is None:  # (an engine nothing was added to yet has an empty z-order, engine.py:664-667)
      return list(self._sprites_and_drapes.keys())
    return self._template.thing_chars()

  @property
  def backdrop(self):
    return self._backdrop

  @property
  def things(self):
    """char -> entity.  Before showtime: the template objects.  After: live
    read-only views of device state (`position`, `visible`, `curtain`...)."""
    if not self._showtime:
      return dict(self._sprites_and_drapes)
    t = self._template
    out = {}
    for i, s in enumerate(t.sprites):
      out[chr(s['ch'])] = _SpriteView(self, i, chr(s['ch']))
    for i, d in enumerate(t.drapes):
      out[chr(d['ch'])] = _DrapeView(self, i, chr(d['ch']))
    return out

  def _read_things(self, sprites=True, curtains=True):
    """Entity state of every environment, only what is asked for: sprites as a
    structured NumPy array [B, n_sprites] (fields row, col, vrow, vcol,
    visible), curtains as uint8 [B, n_drapes, rows, cols]."""
    t = self._template
    B, ns, nd = self._batch, len(t.sprites), len(t.drapes)
    sp = np.zeros((B, max(ns, 1)), _SPRITE_DTYPE) if sprites else None
    cu = np.zeros((B, max(nd, 1), self._rows, self._cols), np.uint8) if curtains else None
    dev.synchronize(self._device_id)
    N.check(N.lib().pcx_engine_read_things(
        self._native, 0, B, sp.ctypes.data if (sprites and ns) else None,
        cu.ctypes.data if (curtains and nd) else None))
    return sp, cu

  @property
  def _b(self):
    """The engine's device buffers; raises once the engine was closed."""
    if not self._bufs:
      raise RuntimeError('this Engine is not in play (its_showtime() has not run, or close() has)')
    return self._bufs

  def close(self):
    """Destroys the device engine and lets go of its device memory (tensors the
    caller still holds -- observations, cropped windows -- stay valid)."""
    if self._native is not None:
      for cropper in self._croppers:  # their native halves point into this engine: they go first
        cropper._release()
      if self._epilogue is not None:  # the kernel stops writing the converter's tensor before anybody frees it
        self._clear_epilogue()
      N.lib().pcx_engine_destroy(self._native)
      self._native = None
      # engine <-> cropper references form a cycle: without this the buffers would wait for the garbage collector
      self._croppers = []
      self._bufs, self._packed, self._actions, self._keepalive = {}, None, None, None

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  # ------------------------------------------------------------------- checks
  def _forbid_after_showtime(self, method_name):
    if self._showtime:
      raise RuntimeError('{} should not be called after its_showtime() '
                         'has been called'.format(method_name))

  def _require_unclaimed(self, characters):
    for char in characters:
      if self._backdrop and char in self._backdrop.palette:
        raise RuntimeError('Character {} is already being used by '
                           'the backdrop'.format(repr(char)))
      if char in self._sprites_and_drapes:
        raise RuntimeError('Character {} is already being used by a sprite '
                           'or a drape'.format(repr(char)))

  def _require_good_characters(self, characters, mandatory_len=None):
    if mandatory_len is not None and len(characters) != mandatory_len:
      raise ValueError(
          '{}, a string of length {}, was used where a string of length {} was '
          'required'.format(repr(characters), len(characters), mandatory_len))
    for char in characters:
      try:
        ord(char)
      except TypeError:
        raise ValueError('Character {} is not an ASCII character'.format(char))


# host image of pcx_sprite_state (include/pcx.h)
_SPRITE_DTYPE = np.dtype([('row', np.int32), ('col', np.int32), ('vrow', np.int32), ('vcol', np.int32),
                          ('visible', np.uint8), ('pad', np.uint8, (3,))])
assert _SPRITE_DTYPE.itemsize == ctypes.sizeof(N.SpriteState)


class _SpriteView(object):
  """Read-only live view of one sprite across the batch.  Batch 1 returns the
  reference's types (`Position`, bool); batch > 1 returns arrays: positions
  int32 [B, 2] (row, col), visibility bool [B] -- one readback of the sprite
  words, no curtains, no Python loop over the batch."""

  def __init__(self, eng, index, character):
    self._eng, self._index, self.character = eng, index, character

  def _state(self):
    sprites, _ = self._eng._read_things(sprites=True, curtains=False)
    return sprites[:, self._index]

  def _positions(self, row, col):
    st = self._state()
    if self._eng.batch == 1:
      return things.Sprite.Position(int(st[row][0]), int(st[col][0]))
    return np.stack([st[row], st[col]], axis=1)

  @property
  def position(self):
    return self._positions('row', 'col')

  @property
  def virtual_position(self):
    return self._positions('vrow', 'vcol')

  @property
  def visible(self):
    vis = self._state()['visible'].astype(np.bool_)
    return bool(vis[0]) if self._eng.batch == 1 else vis


class _DrapeView(object):
  """Read-only live view of one drape's curtain across the batch."""

  def __init__(self, eng, index, character):
    self._eng, self._index, self.character = eng, index, character

  @property
  def curtain(self):
    _, curtains = self._eng._read_things(sprites=False, curtains=True)
    c = curtains[:, self._index].astype(np.bool_)
    return c[0] if self._eng.batch == 1 else c


_ALIAS_TABLE = """
` backtick backquote grave | ~ tilde | 0 zero | 1 one | 2 two | 3 three |
4 four | 5 five | 6 six | 7 seven | 8 eight | 9 nine |
! bang exclamation exclamation_point exclamation_pt | @ at |
# hash hashtag octothorpe number_sign pigpen pound |
$ dollar dollar_sign buck mammon | % percent percent_sign food |
^ carat circumflex trap | & and_sign ampersand | * asterisk star splat |
( lbracket left_bracket lparen left_paren |
) rbracket right_bracket rparen right_paren | - dash hyphen | _ underscore |
+ plus add | = equal equals | [ lsquare left_square_bracket |
] rsquare right_square_bracket |
{ lbrace lcurly left_brace left_curly left_curly_brace |
} rbrace rcurly right_brace right_curly right_curly_brace | PIPE pipe bar |
\\ backslash back_slash reverse_solidus | ; semicolon | : colon |
' tick quote inverted_comma prime |
" quotes double_inverted_commas quotation_mark | z zed | , comma |
< less_than langle left_angle left_angle_bracket | . period full_stop |
> greater_than rangle right_angle right_angle_bracket |
? question question_mark | / slash solidus
"""


def _parse_aliases():
  table = {}
  for entry in _ALIAS_TABLE.replace('\n', ' ').split(' | '):
    words = entry.split()
    if not words:
      continue
    char = '|' if words[0] == 'PIPE' else words[0]
    for name in words[1:]:
      table[name] = char
  return table


class Palette(object):
  """char -> ord helper with attribute aliases (engine.py:877-986)."""

  _ALIASES = _parse_aliases()

  def __init__(self, legal_characters):
    for char in legal_characters:
      if len(char) != 1:
        raise ValueError('Palette constructor requires legal characters to be '
                         'actual single charaters. "{}" is not.'.format(char))
    self._legal_characters = set(legal_characters)

  def __getattr__(self, name):
    if name.startswith('__'):  # keep pickling/copy protocols sane
      raise AttributeError(name)
    return self._lookup(name, AttributeError)

  def __getitem__(self, key):
    return self._lookup(key, IndexError)

  def __getstate__(self):
    return self._legal_characters

  def __setstate__(self, state):
    self._legal_characters = set(state)

  def __contains__(self, key):
    return key in self._legal_characters

  def __iter__(self):
    return iter(self._legal_characters)

  def _lookup(self, key, error):
    key = self._ALIASES.get(key, key)
    if key in self._legal_characters:
      return ord(key)
    raise error('{} is not a legal character in this Palette; legal characters '
                'are {}.'.format(key, list(self._legal_characters)))
