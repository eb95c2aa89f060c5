"""Observation croppers (reference: pycolab/cropping.py:30-598).

Same classes, constructor arguments, validation and error types as the
reference.  `crop()` runs on the device: a cropper owns one window per
environment of the engine's batch (`csrc/pcx_crop.hip`), and the returned
`Observation` holds [rows, cols] NumPy arrays for batch 1 or, for batch > 1,
[B, rows, cols] device tensors that are zero-copy views of the cropper's
output planes -- nothing is copied to the host and nothing waits for the
device; valid until the next `crop()`, like the reference's pre-allocated
output (cropping.py:131-134).
"""

import copy
import ctypes

import numpy as np

from pycolab_amd import _native as N
from pycolab_amd import device as dev
from pycolab_amd import rendering


class ObservationCropper(object):
  """The identity cropper (cropping.py:30-227)."""

  def __init__(self):
    self._engine = None
    self._native = None
    self._pad_char = None
    self._out = None        # output planes [B, 1 + n_chars, pitch] (a device tensor with PyTorch)
    self._pitch = 0
    self._generation = 0    # bumped whenever the native cropper (and its buffers) is rebuilt
    self._fused = False     # the engine's step kernel moves the window and writes the planes (fuse_croppers)
    self._feat_skip_since = 0
    self._feat_skip = 0     # ... and which of the window's own uint8 planes the kernel then no longer writes (1 layers, 2 board too)
    self._features = None   # (converter, float tensor): the window's feature stack the step kernel writes too
                            # (rendering.ObservationToFeatureArray.fuse_into(engine, source=cropper)); the cropper
                            # holds the tensor for as long as the kernel may write it

  def set_engine(self, engine):
    if engine is not self._engine:
      self._release()
      forget = getattr(self._engine, '_unregister_cropper', None)
      if forget is not None:  # (set_engine(None) detaches the cropper)
        forget(self)
    self._engine = engine
    register = getattr(engine, '_register_cropper', None)
    if register is not None:  # lets its_showtime() attach device croppers before frame 0
      register(self)

  def crop(self, observation):
    return observation

  def _clone(self):
    """A cropper with the same constructor arguments and no engine: what a
    batched Story gives every chapter when one cropper object was supplied for
    all of them (a device cropper holds ONE engine's window state)."""
    other = copy.copy(self)
    other._engine, other._native, other._out, other._fused = None, None, None, False
    other._features = None
    other._feat_skip = 0
    other._feat_skip_since = 0
    other._generation = 0
    return other

  @property
  def rows(self):
    return self._engine.rows

  @property
  def cols(self):
    return self._engine.cols

  # -- device plumbing shared by the real croppers -------------------------------
  def _describe(self):
    raise NotImplementedError

  def _release(self):
    if self._native is not None:
      N.lib().pcx_cropper_destroy(self._native)  # (a fused cropper leaves its engine's step kernel first)
      self._native = None
      self._out = None
      self._fused = False
      if self._features is not None:
        self._features[0]._window_gone()
        self._features = None
      self._feat_skip = 0
      self._generation += 1

  def __del__(self):
    try:
      self._release()
    except Exception:  # pylint: disable=broad-except
      pass

  def _device_crop(self):
    """Launches the window update and the copy on the current stream and returns
    the cropped observation: device tensors (zero-copy views of the cropper's
    output planes, no synchronisation) for batch > 1, NumPy arrays of the
    reference's shapes for batch 1."""
    eng = self._engine
    if eng is None or eng._native is None:
      raise RuntimeError('crop() needs set_engine() with an engine that is in play')
    if self._native is None:
      self._create_native()
    lib = N.lib()
    N.check(lib.pcx_cropper_crop(self._native, dev.current_stream(eng._device_id)))
    chars = eng.template.chars
    B, P, r, c = eng.batch, 1 + len(chars), self._rows, self._cols
    if B == 1 or self._out is None or self._out.tensor is None:
      self.check_errors()  # synchronises; batch 1 raises at once, as the reference does (cropping.py:175-183)
      if self._out is not None:
        host = self._out.numpy()
      else:
        planes = ctypes.c_void_p()
        N.check(lib.pcx_cropper_buffers(self._native, ctypes.byref(planes), None))
        host = np.empty((B, P, self._pitch), np.uint8)
        N.check(lib.pcx_memcpy_d2h(host.ctypes.data, planes, host.nbytes))
      host = np.ascontiguousarray(host[:, :, :r * c]).reshape(B, P, r, c)
      if B == 1:
        obs = rendering.Observation(
            board=host[0, 0], layers={chr(ch): host[0, 1 + k].astype(np.bool_) for k, ch in enumerate(chars)})
      else:
        obs = rendering.Observation(
            board=host[:, 0], layers={chr(ch): host[:, 1 + k] for k, ch in enumerate(chars)})
    else:
      seen = N.c_i32(0)
      N.check(lib.pcx_cropper_error_poll(self._native, dev.current_stream(eng._device_id), ctypes.byref(seen)))
      if seen.value:  # an earlier crop() left the observation without a pad character
        self.check_errors()
      view = self._out.tensor.as_strided((B, P, r, c), (P * self._pitch, self._pitch, c, 1))
      obs = rendering.Observation(board=view[:, 0], layers={chr(ch): view[:, 1 + k] for k, ch in enumerate(chars)})
    obs._source = self
    if self._fused and self._feat_skip and eng._steps_launched > self._feat_skip_since:  # (until a step has run, the planes are what the last launch wrote)
      # the step kernel writes this window's float32 stack INSTEAD of (some of) its uint8 planes (fuse_into(...,
      # source=cropper, skip_layers / skip_board)): what is left here is frozen, so it is not handed out -- as
      # Engine._result() does for the full board -- and a post-processor that would read it raises
      obs = rendering.Observation(board=None if self._feat_skip == 2 else obs.board, layers={})
      obs._source = self
      obs._planes_stale = True
    return obs

  def check_errors(self):
    """Synchronises and raises where the reference raises: the window left the
    observation and there is no pad character (cropping.py:175-183).  With
    batch > 1 `crop()` itself does not wait for the device; it reports such an
    error at a later call, or here."""
    if self._native is None:
      return
    errs = np.empty((self._engine.batch,), np.uint8)
    N.check(N.lib().pcx_cropper_errors(self._native, errs.ctypes.data))
    if errs.any():
      raise RuntimeError(
          'An ObservationCropper attempted to crop a region that extends '
          'beyond the observation without specifying a character to fill the '
          'void that exists out there.')

  def _create_native(self):
    eng = self._engine
    if self._native is None and type(self) is not ObservationCropper:
      valid = set(eng.z_order) | {chr(c) for c in eng.template.chars}
      if self._pad_char is not None and self._pad_char not in valid:
        raise ValueError(
            'An `ObservationCropper` tried to fill empty space with a character '
            'that isn\'t used by the current game engine.')
      desc = self._describe()
      handle = ctypes.c_void_p()
      lib = N.lib()
      N.check(lib.pcx_cropper_create(eng._native, ctypes.byref(desc), ctypes.byref(handle)))
      self._native = handle
      self._generation += 1
      self._pitch = int(lib.pcx_cropper_plane_pitch(handle))
      self._out = None
      if dev.torch_module() is not None:  # the output planes are a tensor of ours: crop() hands out views of it
        self._out = dev.DeviceBuffer((eng.batch, 1 + len(eng.template.chars), self._pitch), np.uint8, eng._device_id)
        N.check(lib.pcx_cropper_bind_output(handle, self._out.ptr))

  def _planes_view(self):
    view = N.PlanesView()
    N.check(N.lib().pcx_cropper_planes_view(self._native, ctypes.byref(view)))
    return view, self._engine._device_id


class FixedCropper(ObservationCropper):
  """A constant window (cropping.py:230-268)."""

  def __init__(self, top_left_corner, rows, cols, pad_char=None):
    super(FixedCropper, self).__init__()
    (self._top_row, self._left_col), (self._rows, self._cols) = top_left_corner, (rows, cols)
    self._pad_char = pad_char

  def crop(self, observation):
    del observation  # the device crops the engine's current observation
    return self._device_crop()

  def _describe(self):
    d = N.CropperDesc()
    d.kind, d.rows, d.cols = N.CROP_FIXED, self._rows, self._cols
    d.top, d.left = self._top_row, self._left_col
    d.pad_char = -1 if self._pad_char is None else ord(self._pad_char)
    return d

  @property
  def rows(self):
    return self._rows

  @property
  def cols(self):
    return self._cols


class ScrollingCropper(ObservationCropper):
  """A window that follows game entities (cropping.py:271-598)."""

  def __init__(self, rows, cols, to_track, pad_char=None,
               scroll_margins=(2, 3), initial_offset=None, saccade=True):
    super(ScrollingCropper, self).__init__()
    self._rows, self._cols = rows, cols
    self._to_track = copy.copy(to_track)  # (the caller's list may change later: cropping.py:336)
    self._pad_char = pad_char
    # Margins per axis (cropping.py:338-357): None = keep the entity on the window's centre line, which an even
    # window does not have; a margin that reaches the centre leaves no room to stand in.
    window, margins = (rows, cols), []
    for size, margin in zip(window, scroll_margins):
      if margin is None and size % 2 == 0:
        raise ValueError(
            'A ScrollingCropper can\'t perform perfectly-egocentric scrolling '
            'with a window that has an even number of rows or columns. Either '
            'specify looser scroll margins or use a window with odd dimensions.')
      margins.append(size // 2 if margin is None else margin)
    if any(2 * margin >= size for size, margin in zip(window, margins)):
      raise ValueError(
          'A ScrollingCropper can\'t use scroll margins which extend to or '
          'beyond the very centre of the scrolling window. (Note that if you '
          'haven\'t specified scroll margins and your window is very small or '
          'thin, the default scroll_margins argument might be too big!)')
    self._scroll_margins = tuple(margins)
    self._initial_offset = (0, 0) if initial_offset is None else initial_offset
    self._saccade = saccade

  def set_engine(self, engine):
    prior = self._engine
    super(ScrollingCropper, self).set_engine(engine)
    if engine is not prior and engine is not None:
      if ((engine.rows < self._rows or engine.cols < self._cols)
          and self._pad_char is None):
        raise ValueError(
            'A ScrollingCropper with a size of {} and no pad character '
            'can\'t be used with a pycolab engine that produces smaller '
            'observations in any dimension (in this case, {})'.format(
                (self._rows, self._cols), (engine.rows, engine.cols)))

  def crop(self, observation):
    del observation
    return self._device_crop()

  def _describe(self):
    things = set(self._engine.z_order)
    for entity in self._to_track:
      if entity not in things:
        raise RuntimeError('ScrollingCropper was told to track a nonexistent game entity '
                           '{!r}.'.format(entity))
    d = N.CropperDesc()
    d.kind, d.rows, d.cols = N.CROP_SCROLLING, self._rows, self._cols
    d.pad_char = -1 if self._pad_char is None else ord(self._pad_char)
    d.n_track = len(self._to_track)
    for i, ch in enumerate(self._to_track):
      d.to_track[i] = ord(ch)
    d.margin_rows, d.margin_cols = self._scroll_margins
    d.initial_offset_rows, d.initial_offset_cols = self._initial_offset
    d.saccade = int(bool(self._saccade))
    return d

  @property
  def rows(self):
    return self._rows

  @property
  def cols(self):
    return self._cols


def fuse_croppers(engine, croppers, only_crops=False):
  """Have `engine`'s step kernel run these croppers itself.

  The step kernel holds the frame it paints in LDS, so it can cut the windows
  from it directly: from now on every `its_showtime()` / `play()` / `step()`
  also moves the croppers' windows (one `crop()` per observation, as
  `human_ui.py:269-293` does) and writes their planes in the same launch;
  `cropper.crop(observation)` then launches nothing and returns the window the
  step already wrote.  `only_crops=True` also stops the step from writing the
  full-board planes (the engine's `Observation` goes stale) for consumers that
  only ingest the windows -- on a 45x89 board with a 10x30 egocentric window
  that is 13x less to write.

  Returns False, and changes nothing, where the engine's kernel cannot do it
  (more than four croppers, unoccluded layers, a cropper that tracks a drape
  on a board wider than 128 cells or taller than 63): the croppers then run as their
  own kernels, as before.  `croppers=[]` releases
  them again."""
  croppers = list(croppers)
  for cr in croppers:
    if type(cr) is ObservationCropper:
      raise ValueError('the identity cropper has nothing to fuse')
    cr.set_engine(engine)
  if only_crops:
    others = [cr for cr in getattr(engine, '_croppers', []) if cr not in croppers and type(cr) is not ObservationCropper]
    if others:
      raise ValueError('only_crops=True stops the engine from writing the observation that its %d other cropper(s) '
                       'read: fuse them too, or detach them' % len(others))
  if engine._native is None:  # not in play yet: its_showtime() fuses before frame 0
    engine._fuse_request = (croppers, bool(only_crops))
    return None
  for cr in croppers:
    if cr._native is None:
      cr._create_native()
  arr = (ctypes.c_void_p * max(1, len(croppers)))(*[cr._native for cr in croppers])
  try:
    N.check(N.lib().pcx_engine_fuse_croppers(engine._native, arr, len(croppers), int(bool(only_crops)),
                                             dev.current_stream(engine._device_id)))
  except NotImplementedError:
    return False
  for cr in getattr(engine, '_croppers', []):
    cr._fused = False
  for cr in croppers:
    cr._fused = True
  engine._only_crops = bool(only_crops) and bool(croppers)
  return True


def cropper_from_spec(spec):
  """Build a cropper from the dict form used by the golden fixtures."""
  if spec['kind'] == 'fixed':
    return FixedCropper(tuple(spec['top_left']), spec['rows'], spec['cols'], spec['pad_char'])
  return ScrollingCropper(
      spec['rows'], spec['cols'], list(spec['to_track']), pad_char=spec['pad_char'],
      scroll_margins=tuple(spec['scroll_margins']),
      initial_offset=None if spec['initial_offset'] is None else tuple(spec['initial_offset']),
      saccade=spec['saccade'])
