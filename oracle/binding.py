"""TEST INFRASTRUCTURE: ctypes binding of oracle/liboracle.so.

`OracleEngine` mirrors the batched surface of `pycolab_amd.engine.Engine`
(reset/step/buffers) on host memory so parity tests can feed both the same
template and actions.  Never imported by the product package.
"""
import ctypes
import os
import subprocess

import numpy as np

from pycolab_amd import _native as N

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'liboracle.so')

_SYMS = [
    ('pcxo_engine_create', N.c_i32, [ctypes.POINTER(N.Template), N.c_i64, ctypes.POINTER(ctypes.c_void_p)]),
    ('pcxo_engine_destroy', None, [ctypes.c_void_p]),
    ('pcxo_engine_reset', N.c_i32, [ctypes.c_void_p, ctypes.c_void_p]),
    ('pcxo_engine_step', N.c_i32, [ctypes.c_void_p, ctypes.c_void_p, N.c_i32]),
    ('pcxo_engine_step_hashed', N.c_i32, [ctypes.c_void_p, N.c_u64, N.c_i64, N.c_i64, N.c_i32, N.c_i32]),
    ('pcxo_engine_buffers', N.c_i32, [ctypes.c_void_p, ctypes.POINTER(N.Buffers)]),
    ('pcxo_engine_read_things', N.c_i32, [ctypes.c_void_p, N.c_i64, N.c_i64, ctypes.c_void_p, ctypes.c_void_p]),
    ('pcxo_engine_next_chapter', N.c_i32, [ctypes.c_void_p, ctypes.c_void_p]),
    ('pcxo_engine_plot_words', N.c_i32, [ctypes.c_void_p, ctypes.c_void_p]),
    ('pcxo_engine_set_plot_words', N.c_i32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    ('pcxo_action_hash', N.c_u32, [N.c_u64, N.c_u64, N.c_u64]),
    ('pcxo_last_error', ctypes.c_char_p, []),
    ('pcxo_cropper_create', N.c_i32, [ctypes.c_void_p, ctypes.POINTER(N.CropperDesc), ctypes.POINTER(ctypes.c_void_p)]),
    ('pcxo_cropper_destroy', None, [ctypes.c_void_p]),
    ('pcxo_cropper_crop', N.c_i32, [ctypes.c_void_p]),
    ('pcxo_cropper_buffers', N.c_i32, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p)]),
    ('pcxo_cropper_errors', ctypes.c_void_p, [ctypes.c_void_p]),
]

_lib = None


def build():
  subprocess.check_call(['make', '-s', '-C', HERE])


def lib():
  global _lib
  if _lib is None:
    if not os.path.exists(LIB):
      build()
    _lib = N.bind(ctypes.CDLL(LIB), _SYMS)
  return _lib


def _check(code):
  if code != 0:
    raise RuntimeError('oracle error %d: %s' % (code, lib().pcxo_last_error().decode()))


def action_hash(seed, env, t):
  return lib().pcxo_action_hash(seed, env, t)


def action_hash_np(seed, env, t):
  """Vectorised numpy twin of pcx_action_hash (uint64 arithmetic wraps)."""
  with np.errstate(over='ignore'):
    env = np.asarray(env, dtype=np.uint64)
    t = np.asarray(t, dtype=np.uint64)
    x = (np.uint64(seed) ^ (env * np.uint64(0x9E3779B97F4A7C15))
         ^ (t * np.uint64(0xBF58476D1CE4E5B9)))
    x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return (x >> np.uint64(32)).astype(np.uint32)


class OracleEngine(object):

  def __init__(self, template, batch):
    self.template = template
    self.batch = int(batch)
    ct, self._keep = template.to_ctypes()
    self._h = ctypes.c_void_p()
    _check(lib().pcxo_engine_create(ctypes.byref(ct), self.batch, ctypes.byref(self._h)))
    b = N.Buffers()
    _check(lib().pcxo_engine_buffers(self._h, ctypes.byref(b)))
    B, L, R, C = self.batch, len(template.chars), template.rows, template.cols

    def view(ptr, shape, dtype):
      n = int(np.prod(shape)) * np.dtype(dtype).itemsize
      buf = (ctypes.c_uint8 * n).from_address(ptr)
      return np.frombuffer(buf, dtype=dtype).reshape(shape)

    self.planes = view(b.planes, (B, 1 + L, R, C), np.uint8)
    self.reward = view(b.reward, (B,), np.float32 if getattr(template, 'reward_is_float', False) else np.int32)
    self.reward_set = view(b.reward_set, (B,), np.uint8)
    self.discount = view(b.discount, (B,), np.float32)
    self.done = view(b.done, (B,), np.uint8)
    self.frame = view(b.frame, (B,), np.int32)
    self.error = view(b.error, (B,), np.uint8)

  def reset(self, mask=None):
    ptr = None
    if mask is not None:
      mask = np.ascontiguousarray(mask, np.uint8)
      ptr = mask.ctypes.data
    _check(lib().pcxo_engine_reset(self._h, ptr))

  def step(self, actions, auto_reset=True):
    actions = np.ascontiguousarray(actions, np.int32).reshape(self.batch)
    _check(lib().pcxo_engine_step(self._h, actions.ctypes.data, int(auto_reset)))

  def step_hashed(self, seed, t0, steps, env_offset=0, auto_reset=True):
    _check(lib().pcxo_engine_step_hashed(self._h, seed, env_offset, t0, steps, int(auto_reset)))

  def next_chapter(self):
    """int32 [batch]: what the entities assigned to the_plot.next_chapter (N.CHAPTER_NONE / N.CHAPTER_UNSET)."""
    out = np.zeros((self.batch,), np.int32)
    _check(lib().pcxo_engine_next_chapter(self._h, out.ctypes.data))
    return out

  def plot_words(self):
    """int32 [PLOT_WORDS, batch] (include/pcx.h pcx_engine_plot_words)."""
    out = np.zeros((N.PLOT_WORDS, self.batch), np.int32)
    _check(lib().pcxo_engine_plot_words(self._h, out.ctypes.data))
    return out

  def set_plot_words(self, words, mask=None):
    words = np.ascontiguousarray(words, np.int32).reshape(N.PLOT_WORDS, self.batch)
    ptr = None
    if mask is not None:
      mask = np.ascontiguousarray(mask, np.uint8)
      ptr = mask.ctypes.data
    _check(lib().pcxo_engine_set_plot_words(self._h, words.ctypes.data, ptr))

  def sprites(self):
    ns = len(self.template.sprites)
    arr = (N.SpriteState * (self.batch * max(ns, 1)))()
    _check(lib().pcxo_engine_read_things(self._h, 0, self.batch, ctypes.addressof(arr), None))
    out = np.zeros((self.batch, ns, 5), np.int16)
    for b in range(self.batch):
      for s in range(ns):
        st = arr[b * ns + s]
        out[b, s] = (st.row, st.col, st.vrow, st.vcol, st.visible)
    return out

  def curtains(self):
    nd = len(self.template.drapes)
    out = np.zeros((self.batch, nd, self.template.rows, self.template.cols), np.uint8)
    _check(lib().pcxo_engine_read_things(self._h, 0, self.batch, None, out.ctypes.data))
    return out

  def close(self):
    if self._h:
      lib().pcxo_engine_destroy(self._h)
      self._h = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass


class OracleCropper(object):
  """One cropper (a `pycolab_amd.cropping` object supplies the descriptor)
  over an OracleEngine's batch."""

  def __init__(self, engine, cropper):
    cropper._engine = _EngineShim(engine)
    desc = cropper._describe()
    self.engine, self.rows, self.cols = engine, desc.rows, desc.cols
    self._h = ctypes.c_void_p()
    _check(lib().pcxo_cropper_create(engine._h, ctypes.byref(desc), ctypes.byref(self._h)))

  def crop(self):
    _check(lib().pcxo_cropper_crop(self._h))
    planes, corner = ctypes.c_void_p(), ctypes.c_void_p()
    _check(lib().pcxo_cropper_buffers(self._h, ctypes.byref(planes), ctypes.byref(corner)))
    B, P = self.engine.batch, 1 + len(self.engine.template.chars)
    n = B * P * self.rows * self.cols
    arr = np.frombuffer((ctypes.c_uint8 * n).from_address(planes.value), np.uint8).reshape(B, P, self.rows, self.cols)
    err = np.frombuffer((ctypes.c_uint8 * B).from_address(lib().pcxo_cropper_errors(self._h)), np.uint8)
    return arr.copy(), err.copy()

  def close(self):
    if self._h:
      lib().pcxo_cropper_destroy(self._h)
      self._h = None


class _EngineShim(object):
  """What a cropper's _describe() looks at."""

  def __init__(self, engine):
    self.z_order = engine.template.thing_chars()
    self.rows, self.cols = engine.template.rows, engine.template.cols
