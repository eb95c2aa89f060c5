"""Observations and observation post-processors.

`Observation` is the reference's namedtuple (rendering.py:28-63).  In batched
mode `board` has shape [B, rows, cols] and every `layers[c]` has shape
[B, rows, cols]; all are zero-copy views of a planes array in HBM
([B, 1 + n_chars, pitch] uint8), valid until the next step -- the
reference's own aliasing rule.

`ObservationToArray`, `ObservationToFeatureArray` and
`ObservationCharacterRepainter` keep the reference's constructors
(rendering.py:304-661) and run as streaming epilogue kernels
(csrc/pcx_post.hip) over the planes the engine, a cropper or a repainter
just wrote.  With batch > 1 their results are device tensors the kernel wrote
in place (no copy, no synchronisation; `torch.utils.dlpack.to_dlpack(t)`
exports them to any DLPack consumer); batch 1 returns the reference's NumPy
types.  They accept observations produced by this package (those know where
their planes live); anything else is rejected -- there is no host path.
"""

import collections
import ctypes

import numpy as np

from pycolab_amd import _native as N
from pycolab_amd import device as dev


class Observation(collections.namedtuple('Observation', ['board', 'layers'])):
  """board + layers; instances made by engines/croppers remember their source."""
  _source = None


_DTYPES = {'uint8': N.U8, 'int32': N.I32, 'float32': N.F32, 'int64': N.I64, 'float64': N.F64}


class _Post(object):
  """One device post-processor bound to one planes source.

  The output array is a device tensor of ours that the C side writes into
  (`pcx_post_bind_output`): `run()` enqueues one kernel on the current stream
  and returns that tensor -- no copy and no wait (batch 1, or no PyTorch: a
  NumPy copy, after a synchronisation, in the reference's shapes).
  """

  def __init__(self, source, desc, out_dtype, out_shape):
    self.view, self.device_id = source._planes_view()
    self.key = _view_key(self.view)
    self.desc = desc
    self.out_dtype, self.out_shape = np.dtype(out_dtype), tuple(out_shape)
    handle = ctypes.c_void_p()
    lib = N.lib()
    N.check(lib.pcx_post_create(ctypes.byref(self.view), ctypes.byref(desc), self.device_id, ctypes.byref(handle)))
    self.handle = handle
    self.out = None
    if dev.torch_module() is not None:
      self.out = dev.DeviceBuffer(self.out_shape, self.out_dtype, self.device_id)
      N.check(lib.pcx_post_bind_output(handle, self.out.ptr, self.out.nbytes))

  def run(self, host):
    lib = N.lib()
    N.check(lib.pcx_post_run(self.handle, dev.current_stream(self.device_id)))
    if self.out is not None and self.out.tensor is not None and not host:
      seen = N.c_i32(0)
      N.check(lib.pcx_post_error_poll(self.handle, dev.current_stream(self.device_id), ctypes.byref(seen)))
      return self.out.tensor, bool(seen.value)
    dev.synchronize(self.device_id)
    if self.out is not None:
      return self.out.numpy(), self.errors().any()
    ptr, nbytes = ctypes.c_void_p(), N.c_u64()
    N.check(lib.pcx_post_output(self.handle, ctypes.byref(ptr), ctypes.byref(nbytes)))
    out = np.empty(self.out_shape, self.out_dtype)
    assert out.nbytes == nbytes.value, (out.nbytes, nbytes.value)
    N.check(lib.pcx_memcpy_d2h(out.ctypes.data, ptr, out.nbytes))
    return out, self.errors().any()

  def errors(self):
    errs = np.empty((int(self.view.batch),), np.uint8)
    N.check(N.lib().pcx_post_errors(self.handle, errs.ctypes.data))
    return errs

  def __del__(self):
    try:
      N.lib().pcx_post_destroy(self.handle)
    except Exception:  # pylint: disable=broad-except
      pass


def _view_key(view):
  """What a cached post-processor depends on: where the source planes live and
  their shape.  A cropper re-attached to another engine, or an engine that was
  closed and rebuilt, shows up as a different key (the old planes are gone)."""
  return (view.planes, int(view.batch), view.rows, view.cols, view.pitch, view.n_chars, bytes(view.chars))


def _bound(post, source):
  """The cached `_Post` if it still reads the planes `source` has now."""
  if post is None:
    return None
  view, _ = source._planes_view()
  return post if post.key == _view_key(view) else None


def _source_of(observation):
  source = getattr(observation, '_source', None)
  if source is None:
    raise TypeError('this post-processor runs on the device and needs an Observation returned by a '
                    'pycolab_amd Engine or cropper')
  if getattr(observation, '_planes_stale', False):
    raise RuntimeError('this Observation is a placeholder: its engine no longer writes the full-board planes '
                       '(fuse_croppers(..., only_crops=True) or fuse_into(..., skip_board=True)); only the fused '
                       'croppers / the fused converter have something to return')
  return source


# Where the general epilogue body loses to step + stand-alone kernel (profiles/r04_post_kernels.md: scrolly_maze at
# 1,048,576 environments, value array next to the planes 1.33 against 0.95 ms, repainter 1.94 against 1.42; hello_world
# at 262,144: 0.40 / 0.36 and 0.64 / 0.51) and where it wins (4,096: 0.018 / 0.027; marauders 32,768: 0.059 / 0.071):
# the dividing line is whether the step's own planes still fit the 256 MiB Infinity Cache -- beyond it the separate
# kernel streams the board back at cache speed while the fused body is bound by its scalar bookkeeping.
_FUSED_BODY_PAYS_UP_TO = 256 << 20


def _general_epilogue_pays(engine):
  """Do this engine's observation planes fit the Infinity Cache (the fused value-array / repainter body wins)?"""
  return engine.batch * (1 + len(engine.template.chars)) * engine._pitch <= _FUSED_BODY_PAYS_UP_TO


def _strides(shape3, permute):
  """Element strides of (d, r, c) in the contiguous array whose axes are
  `permute` of (d, r, c) -- np.transpose(result, permute) made contiguous."""
  permute = (0, 1, 2) if permute is None else tuple(permute)
  out_shape = [shape3[a] for a in permute]
  out_strides = [int(np.prod(out_shape[i + 1:])) for i in range(3)]
  stride = [0, 0, 0]
  for out_axis, a in enumerate(permute):
    stride[a] = out_strides[out_axis]
  return stride, out_shape


class ObservationToArray(object):
  """board characters -> values or value vectors (rendering.py:409-542)."""

  def __init__(self, value_mapping, dtype=None, permute=None):
    self._value_mapping = value_mapping
    first = next(iter(value_mapping.values()))
    self._dtype = np.dtype(dtype if dtype is not None else np.array(first).dtype)
    try:
      self._depth = len(first)
      self._is_3d = True
    except TypeError:
      self._depth = 1
      self._is_3d = False
    self._permute = tuple(permute) if permute is not None else None
    if permute is not None:
      if self._is_3d and set(permute) != {0, 1, 2}:
        raise ValueError('When the value mapping contains 1-D vectors, the '
                         'permute argument to the ObservationToArray '
                         'constructor must be a list or tuple containing some '
                         'permutation of the integers 0, 1, and 2.')
      elif not self._is_3d and set(permute) != {0, 1}:
        raise ValueError('When the value mapping contains scalars, the permute '
                         'argument to the ObservationToArray constructor must '
                         'be a list or tuple containing some permutation of '
                         'the integers 0 and 1.')
    if self._dtype.name not in _DTYPES:
      raise NotImplementedError('ObservationToArray on the device supports dtypes {}'.format(sorted(_DTYPES)))
    if self._depth > N.POST_MAX_DEPTH:
      raise NotImplementedError('value vectors longer than {}'.format(N.POST_MAX_DEPTH))
    self._post = None
    self._fused = None  # (engine, device tensor, step count) once fuse_into() succeeded

  def fuse_into(self, engine, skip_layers=False, skip_board=False, force=False):
    """Have `engine`'s step kernel write this array itself, as an epilogue of
    its render loop (the board dword is in a register there, the value table in
    LDS): from the next `play()` / `step()` on, calling this object with one of
    the engine's observations returns the tensor the step already filled.
    `skip_layers=True` also stops the step from writing the uint8 layer planes
    -- a consumer that ingests, say, an RGB image gets the board and the image
    and nothing else; `skip_board=True` leaves out the board plane too (the
    engine's observations then carry `board=None`): the step writes this one
    array (scrolly_maze, a million environments, RGB: 0.69 ms against 1.05 ms
    for step + post-processor).  Returns False, and changes nothing, where the
    engine's kernel cannot do it (batch 1, a permuted axis order, boards that
    are not a whole number of dwords, a character of the game without a value,
    the table-driven kernel, unoccluded layers, fused croppers) -- or where it
    would be SLOWER than the two kernels: with the planes kept (or only the
    layers dropped) on batches whose planes no longer fit the Infinity Cache
    (`_general_epilogue_pays`; `force=True` fuses regardless, for A/B runs).
    Calls then run the post-processor as its own kernel, as before."""
    torch = dev.torch_module()
    identity = self._permute is None or tuple(self._permute) == tuple(range(len(self._permute)))
    if not identity or torch is None or engine._native is None or engine.batch == 1:
      return False
    if not (force or skip_board or _general_epilogue_pays(engine)):
      return False
    tdtype = getattr(torch, self._dtype.name)
    shape = (self._depth, engine.rows, engine.cols) if self._is_3d else (engine.rows, engine.cols)
    out = torch.zeros((engine.batch,) + shape, dtype=tdtype, device='cuda:%d' % engine._device_id)
    lut = np.zeros((N.POST_MAX_DEPTH, 128), np.uint64)
    mapped = np.zeros((128,), np.uint8)
    for key, value in self._value_mapping.items():
      ch = ord(key)
      if ch > 127:
        continue
      mapped[ch] = 1
      comps = np.atleast_1d(np.array(value)).astype(self._dtype)
      for i in range(self._depth):
        lut[i, ch] = int.from_bytes(comps[i].tobytes().ljust(8, b'\0'), 'little')
    d = N.EpilogueDesc()
    d.depth = self._depth
    d.out_dev = out.data_ptr()
    d.skip_layers = 2 if skip_board else int(bool(skip_layers))
    d.to_array = 1
    d.dtype = _DTYPES[self._dtype.name]
    d.lut = lut.ctypes.data
    d.mapped = mapped.ctypes.data
    # start from the current observation (environments a later step leaves untouched keep values that match their
    # board) -- computed BEFORE the kernel's epilogue is switched: if the engine has no planes to start from (an
    # earlier epilogue with skip_board) the array starts as zeros and nothing is left half-installed
    try:
      seed = ObservationToArray(self._value_mapping, self._dtype, self._permute)(engine._result()[0])
    except RuntimeError:
      seed = None
    try:
      N.check(N.lib().pcx_engine_set_epilogue(engine._native, ctypes.byref(d)))
    except NotImplementedError:
      return False
    if seed is not None:
      out.copy_(seed)
    engine._install_epilogue(self, out, only=skip_board)  # the ENGINE owns the epilogue (see ObservationToFeatureArray.fuse_into)
    self._fused = (engine, out, engine._steps_launched)
    return True

  def unfuse(self):
    """Takes the epilogue out of the engine's step kernel again."""
    if self._fused is not None:
      self._fused[0]._clear_epilogue(self)

  def _epilogue_gone(self):
    self._fused = None

  def _after_import(self, engine, out, restored):
    """Engine.import_state(): `out` holds what the EXPORTING engine's kernel wrote.  With the observation restored it
    is refilled from the restored planes; without, it counts as not written until the next step."""
    if restored:
      out.copy_(ObservationToArray(self._value_mapping, self._dtype, self._permute)(engine._result()[0]))
      self._fused = (engine, out, engine._steps_launched - 1)
    else:
      self._fused = (engine, out, engine._steps_launched)

  def __call__(self, observation):
    """Batch 1: a NumPy array as in the reference.  Batch > 1: a device tensor
    [B, ...] written in place by the kernel (no copy, no synchronisation); an
    unmapped character is then reported at a later call or by
    `check_errors()`."""
    if self._fused is not None and getattr(observation, '_source', None) is self._fused[0]:
      engine, out, attached_at = self._fused
      if engine._steps_launched > attached_at:  # a step has run since: the kernel wrote `out`
        return out
    source = _source_of(observation)
    self._post = _bound(self._post, source)
    if self._post is None:
      view, _ = source._planes_view()
      R, C = view.rows, view.cols
      d = N.PostDesc()
      d.kind, d.dtype, d.depth = N.POST_TO_ARRAY, _DTYPES[self._dtype.name], self._depth
      for key, value in self._value_mapping.items():
        ch = ord(key)
        if ch > 127:
          continue
        d.mapped[ch] = 1
        comps = np.atleast_1d(np.array(value)).astype(self._dtype)
        for i in range(self._depth):
          d.lut[i][ch] = int.from_bytes(comps[i].tobytes().ljust(8, b'\0'), 'little')
      if self._is_3d:
        perm3 = self._permute
      else:  # (rows, cols) permutation of the 2-D result
        perm3 = None if self._permute is None else (0,) + tuple(1 + a for a in self._permute)
      stride, shape = _strides((self._depth, R, C), perm3)
      d.stride[0], d.stride[1], d.stride[2] = stride
      if not self._is_3d:
        shape = shape[1:]
      self._post = _Post(source, d, self._dtype, (int(view.batch),) + tuple(shape))
    single = self._post.out_shape[0] == 1
    out, bad = self._post.run(host=single)
    if bad:
      self.check_errors()
    return out[0] if single else out

  def check_errors(self):
    """Synchronises; raises if an observation held a character without a value."""
    if self._post is not None and self._post.errors().any():
      raise RuntimeError(
          'This ObservationToArray only knows array values for the '
          'characters {}, but it received an observation with a character '
          'not in that set'.format(str(''.join(self._value_mapping.keys()))))


class ObservationToFeatureArray(object):
  """float32 stack of selected layers (rendering.py:545-661)."""

  def __init__(self, layers, permute=None):
    self._layers = layers
    self._depth = len(layers)
    self._permute = tuple(permute) if permute is not None else None
    if permute is not None and sorted(permute) != [0, 1, 2]:
      raise ValueError('The permute argument to the ObservationToFeatureArray '
                       'constructor must be a list or tuple containing some '
                       'permutation of the integers 0, 1, and 2.')
    if self._depth > N.POST_MAX_DEPTH:
      raise NotImplementedError('more than {} layers'.format(N.POST_MAX_DEPTH))
    self._post = None
    self._fused = None  # (engine, device tensor) once fuse_into() succeeded
    self._fused_window = None  # (cropper, device tensor, step count): fuse_into(engine, source=cropper)

  def fuse_into(self, engine, skip_layers=False, skip_board=False, source=None):
    """Have `engine`'s step kernel write this feature array itself, as an
    epilogue of its render loop (the layer masks are in registers there):
    from the next `play()` / `step()` on, calling this object with one of the
    engine's observations returns the tensor the step already filled -- no
    second pass over the planes, no extra launch.  `skip_layers=True` also
    stops the step from writing the uint8 layer planes (`Observation.layers`
    then goes stale; the board stays valid) for consumers that only ingest
    the features.  Returns False, and changes nothing, where the engine's
    kernel cannot do it (batch > 1; the five hand-written step kernels with
    occluded layers and no fused croppers; axis order default or channels
    last, `permute=(1, 2, 0)` -- the latter on boards of whole dwords): calls
    then run the post-processor as its own kernel, as before.  The
    engine owns the installed epilogue: when this object goes away, or another
    post-processor fuses, the kernel stops writing into this one's tensor.

    `source=cropper` -- crop, THEN post-process, in the one launch (the order
    of the reference's pipeline: human_ui.py:252-265, better_scrolly_maze.py:
    237-247 into rendering.py:545-661): `cropper` must be fused into the
    engine's step kernel (`cropping.fuse_croppers`); the kernel then also
    writes the feature stack OF ITS WINDOW, [B, depth, rows, cols] of the
    window (channels last: [B, rows, cols, depth]), and calling this object
    with one of the cropper's observations returns that tensor.  Here
    `skip_layers` / `skip_board` stop the kernel from writing the WINDOW's
    uint8 layer planes / board plane (with `fuse_croppers(..., only_crops=
    True)` a learner that ingests the egocentric feature stack gets exactly
    that and nothing else: better_scrolly_maze, 10x30 window of the 45x89
    board, eight layers: 9.6 KB per environment instead of 32 KB of planes
    plus a second pass).  False for the table-driven kernel, more than 16
    layers, a cropper that is not fused."""
    if source is not None:
      return self._fuse_into_window(engine, source, skip_layers, skip_board)
    torch = dev.torch_module()
    if (self._permute not in (None, (0, 1, 2), (1, 2, 0)) or torch is None or engine._native is None or engine.batch == 1 or
        len(set(self._layers)) != len(self._layers) or any(ord(c) > 255 for c in self._layers)):
      return False
    channels_last = self._permute == (1, 2, 0)
    shape = (engine.rows, engine.cols, self._depth) if channels_last else (self._depth, engine.rows, engine.cols)
    out = torch.zeros((engine.batch,) + shape, dtype=torch.float32, device='cuda:%d' % engine._device_id)
    d = N.EpilogueDesc()
    d.depth = self._depth
    for i, ch in enumerate(self._layers):
      d.chars[i] = ord(ch)
    d.out_dev = out.data_ptr()
    d.skip_layers = 2 if skip_board else int(bool(skip_layers))  # skip_board: not even the board plane (observations carry board=None)
    d.channels_last = int(channels_last)
    # start from the current observation (environments a later step leaves untouched keep features that match their
    # planes), computed before the kernel's epilogue is switched (see ObservationToArray.fuse_into)
    try:
      seed = ObservationToFeatureArray(self._layers, self._permute)(engine._result()[0])
    except RuntimeError:
      seed = None
    try:
      N.check(N.lib().pcx_engine_set_epilogue(engine._native, ctypes.byref(d)))
    except NotImplementedError:
      return False
    if seed is not None:
      out.copy_(seed)
    # The ENGINE owns the epilogue: it keeps this converter and the tensor the
    # kernel writes alive for as long as it may launch (a converter that was
    # garbage-collected would leave the kernel writing freed memory), tells the
    # converter it replaces that it is no longer fed, and lets go in close().
    engine._install_epilogue(self, out, only=skip_board)
    self._fused = (engine, out, engine._steps_launched)
    return True

  def unfuse(self):
    """Takes the epilogue out of the engine's step kernel again (calls then run
    the post-processor as its own kernel)."""
    if self._fused is not None:
      self._fused[0]._clear_epilogue(self)
    if self._fused_window is not None:
      cropper = self._fused_window[0]
      if cropper._native is not None:
        N.check(N.lib().pcx_cropper_set_features(cropper._native, None))
      cropper._features = None
      cropper._feat_skip = 0
      self._fused_window = None

  def _epilogue_gone(self):
    self._fused = None

  def _fuse_into_window(self, engine, cropper, skip_layers, skip_board):
    torch = dev.torch_module()
    if (torch is None or engine._native is None or engine.batch == 1 or cropper._engine is not engine or not cropper._fused or
        cropper._native is None or self._permute not in (None, (0, 1, 2), (1, 2, 0)) or
        len(set(self._layers)) != len(self._layers) or any(ord(c) > 255 for c in self._layers) or self._depth > 16):
      return False
    channels_last = self._permute == (1, 2, 0)
    shape = (cropper.rows, cropper.cols, self._depth) if channels_last else (self._depth, cropper.rows, cropper.cols)
    out = torch.zeros((engine.batch,) + shape, dtype=torch.float32, device='cuda:%d' % engine._device_id)
    d = N.EpilogueDesc()
    d.depth = self._depth
    for i, ch in enumerate(self._layers):
      d.chars[i] = ord(ch)
    d.out_dev = out.data_ptr()
    d.skip_layers = 2 if skip_board else int(bool(skip_layers))
    d.channels_last = int(channels_last)
    # (the window as it stands, through the stand-alone kernel: environments a later step leaves alone keep it)
    seed = ObservationToFeatureArray(self._layers, self._permute)(cropper.crop(None))
    try:
      N.check(N.lib().pcx_cropper_set_features(cropper._native, ctypes.byref(d)))
    except NotImplementedError:
      return False
    out.copy_(seed)
    if cropper._features is not None and cropper._features[0] is not self:
      cropper._features[0]._window_gone()
    cropper._features = (self, out)
    cropper._feat_skip = int(d.skip_layers)
    cropper._feat_skip_since = engine._steps_launched
    self._fused_window = (cropper, out, engine._steps_launched)
    return True

  def _window_gone(self):
    self._fused_window = None

  def _window_after_import(self, cropper, out, restored=True):
    """Engine.import_state() with a stack fused into `cropper`'s window: the tensor holds what the step kernel wrote
    before the import.  Refill it from the restored window through the stand-alone kernels (the window's uint8 planes
    are restored with the checkpoint's observation or recut by crop()); a window whose planes the kernel no longer
    writes (skip_layers / skip_board) cannot be recomputed and counts as not written until the next step -- and so does
    every window when the checkpoint carried no observation (`restored` False: crop(None) would recut it from the
    engine's full-board planes, which then still show the frame BEFORE the import)."""
    if self._fused_window is None or self._fused_window[0] is not cropper:
      return
    engine = cropper._engine
    if getattr(cropper, '_feat_skip', 0) or not restored:
      self._fused_window = (cropper, out, engine._steps_launched)  # (stale until a step rewrites it)
      return
    out.copy_(ObservationToFeatureArray(self._layers, self._permute)(cropper.crop(None)))
    self._fused_window = (cropper, out, engine._steps_launched - 1)

  def _after_import(self, engine, out, restored):
    """Engine.import_state(): see ObservationToArray._after_import."""
    if restored:
      out.copy_(ObservationToFeatureArray(self._layers, self._permute)(engine._result()[0]))
      self._fused = (engine, out, engine._steps_launched - 1)
    else:
      self._fused = (engine, out, engine._steps_launched)

  def __call__(self, observation):
    if self._fused_window is not None and getattr(observation, '_source', None) is self._fused_window[0]:
      cropper, out, attached_at = self._fused_window
      if cropper._fused and cropper._engine is not None and cropper._engine._steps_launched > attached_at:
        return out  # the step kernel cut the window and wrote its stack
    if self._fused is not None and getattr(observation, '_source', None) is self._fused[0]:
      engine, out, attached_at = self._fused
      if engine._steps_launched > attached_at:  # a step has run since: the kernel wrote `out`
        return out
    if getattr(observation, '_planes_stale', False):
      _source_of(observation)  # (raises, saying what the placeholder is: BEFORE the layers check, which its empty `layers` would fail with a misleading text)
    if not any(l in observation.layers for l in self._layers):
      raise RuntimeError(
          'The layers argument to this ObservationToFeatureArray, {}, has no '
          'entry that refers to an actual feature in the input observation. '
          'Actual features in the observation are {}.'.format(
              repr(self._layers), repr(''.join(sorted(observation.layers)))))
    source = _source_of(observation)
    self._post = _bound(self._post, source)
    if self._post is None:
      view, _ = source._planes_view()
      d = N.PostDesc()
      d.kind, d.dtype, d.depth = N.POST_FEATURE_ARRAY, N.F32, self._depth
      for i, ch in enumerate(self._layers):
        d.chars[i] = ord(ch) if ord(ch) < 256 else 255
      stride, shape = _strides((self._depth, view.rows, view.cols), self._permute)
      d.stride[0], d.stride[1], d.stride[2] = stride
      self._post = _Post(source, d, np.float32, (int(view.batch),) + tuple(shape))
    single = self._post.out_shape[0] == 1
    out, _ = self._post.run(host=single)  # float32 device tensor [B, ...] for batch > 1 (zero-copy hand-off)
    return out[0] if single else out


class ObservationCharacterRepainter(object):
  """Repaints characters through a fixed mapping (rendering.py:304-406)."""

  def __init__(self, character_mapping):
    self._character_mapping = character_mapping
    self._post = None
    self._out_chars = None
    self._repainted = None
    self._fused = None  # (engine, step count) once fuse_into() succeeded

  def fuse_into(self, engine, skip_layers=False, skip_board=False, force=False):
    """Have `engine`'s step kernel write the repainted observation itself (the
    board dword through the character table, every output layer by a byte-wise
    compare): from the next `play()` / `step()` on, calling this object with
    one of the engine's observations returns the planes the step already
    filled; they are a planes source as before (`ObservationToArray(...)(
    repainter(observation))` chains).  `skip_layers` / `skip_board`: the step
    no longer writes the original layer planes / any original plane.  Returns
    False, and changes nothing, where the engine's kernel cannot do it (batch
    1, boards that are not a whole number of dwords, the table-driven kernel,
    unoccluded layers, fused croppers) or where it would be slower than step +
    stand-alone kernel: batches whose planes no longer fit the Infinity Cache
    (`_general_epilogue_pays`; `force=True` fuses regardless)."""
    torch = dev.torch_module()
    if torch is None or engine._native is None or engine.batch == 1:
      return False
    if not (force or _general_epilogue_pays(engine)):
      return False
    obs = engine._result()[0]
    if obs.board is None:
      return False
    self(obs)  # creates the output planes (and brings them up to date with the current observation)
    if self._post.out is None or self._post.out.tensor is None:
      return False
    lut = np.zeros((N.POST_MAX_DEPTH, 128), np.uint64)
    lut[0, :] = np.arange(128)
    for k, v in self._character_mapping.items():
      lut[0, ord(k)] = ord(v)
    mapped = np.ones((128,), np.uint8)
    d = N.EpilogueDesc()
    d.depth = len(self._out_chars)
    for i, ch in enumerate(self._out_chars):
      d.chars[i] = ord(ch)
    d.out_dev = self._post.out.ptr
    d.skip_layers = 2 if skip_board else int(bool(skip_layers))
    d.to_array = 2
    d.dtype = N.U8
    d.lut = lut.ctypes.data
    d.mapped = mapped.ctypes.data
    try:
      N.check(N.lib().pcx_engine_set_epilogue(engine._native, ctypes.byref(d)))
    except NotImplementedError:
      return False
    engine._install_epilogue(self, self._post.out, only=skip_board)
    self._fused = (engine, engine._steps_launched)
    return True

  def unfuse(self):
    if self._fused is not None:
      self._fused[0]._clear_epilogue(self)

  def _epilogue_gone(self):
    self._fused = None

  def _after_import(self, engine, out, restored):
    """Engine.import_state(): the repainted planes are the exporting engine's; the next call repaints the restored
    observation with the stand-alone kernel (into the same planes), later ones take what the step kernel writes."""
    del out, restored
    self._fused = (engine, engine._steps_launched)

  def __call__(self, original_observation):
    fused_now = (self._fused is not None and getattr(original_observation, '_source', None) is self._fused[0] and
                 self._fused[0]._steps_launched > self._fused[1])
    if fused_now:
      return self._wrap(self._post.out.tensor)  # the step kernel wrote the planes
    source = _source_of(original_observation)
    self._post = _bound(self._post, source)
    if self._post is None:
      view, device_id = source._planes_view()
      self._out_chars = sorted((set(original_observation.layers) - set(self._character_mapping))
                               .union(self._character_mapping.values()))
      if len(self._out_chars) > N.POST_MAX_DEPTH:
        raise NotImplementedError('a repainted observation with more than {} characters'.format(N.POST_MAX_DEPTH))
      d = N.PostDesc()
      d.kind, d.dtype, d.depth = N.POST_REPAINT, N.U8, len(self._out_chars)
      for ch in range(128):
        d.lut[0][ch] = ch
        d.mapped[ch] = 1
      for k, v in self._character_mapping.items():
        if ord(k) > 127 or ord(v) > 127:
          raise ValueError('ObservationCharacterRepainter on the device repaints ASCII characters only')
        d.lut[0][ord(k)] = ord(v)
      for i, ch in enumerate(self._out_chars):
        d.chars[i] = ord(ch)
      pitch = (view.rows * view.cols + 3) & ~3
      self._post = _Post(source, d, np.uint8, (int(view.batch), 1 + len(self._out_chars), pitch))
      self._repainted = _RepaintedPlanes(self._post, self._out_chars, view.rows, view.cols, pitch, device_id)
    B = self._post.out_shape[0]
    planes, bad = self._post.run(host=B == 1)
    if bad and self._post.errors().any():
      raise RuntimeError('ObservationCharacterRepainter met a character outside ASCII')
    return self._wrap(planes)

  def _wrap(self, planes):
    B = self._post.out_shape[0]
    R, C, pitch = self._repainted.rows, self._repainted.cols, self._repainted.pitch
    if B == 1:
      planes = np.ascontiguousarray(planes[0, :, :R * C]).reshape(-1, R, C)
      obs = Observation(board=planes[0],
                        layers={c: planes[1 + i].astype(np.bool_) for i, c in enumerate(self._out_chars)})
    else:
      P = 1 + len(self._out_chars)
      if hasattr(planes, 'as_strided'):   # device tensor: zero-copy [B, P, rows, cols] view of the pitched planes
        view4 = planes.as_strided((B, P, R, C), (P * pitch, pitch, C, 1))
      else:
        view4 = np.ascontiguousarray(planes[:, :, :R * C]).reshape(B, P, R, C)
      obs = Observation(board=view4[:, 0], layers={c: view4[:, 1 + i] for i, c in enumerate(self._out_chars)})
    obs._source = self._repainted  # lets ObservationToArray / ToFeatureArray run on the repainted planes
    return obs


class _RepaintedPlanes(object):
  """The planes a repainter wrote, as a source for further post-processors."""

  def __init__(self, post, out_chars, rows, cols, pitch, device_id):
    self._post, self._out_chars = post, out_chars
    self.rows, self.cols, self.pitch, self._device_id = rows, cols, pitch, device_id

  def _planes_view(self):
    ptr, nbytes = ctypes.c_void_p(), N.c_u64()
    N.check(N.lib().pcx_post_output(self._post.handle, ctypes.byref(ptr), ctypes.byref(nbytes)))
    view = N.PlanesView()
    view.planes = ptr.value
    view.batch = self._post.out_shape[0]
    view.rows, view.cols, view.pitch = self.rows, self.cols, self.pitch
    view.n_chars = len(self._out_chars)
    for i, ch in enumerate(self._out_chars):
      view.chars[i] = ord(ch)
    return view, self._device_id
