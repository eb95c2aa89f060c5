"""ctypes mirror of `include/pcx.h` and the loader of `csrc/libpcx.so`.

The shared library is the product: if it is missing this module raises, there
is no CPU fallback behind it.
"""

import ctypes
import os

c_u8, c_i32, c_i64, c_u32, c_u64 = (ctypes.c_uint8, ctypes.c_int32,
                                    ctypes.c_int64, ctypes.c_uint32,
                                    ctypes.c_uint64)
c_u8_p = ctypes.POINTER(ctypes.c_uint8)

ABI_VERSION = 4
MAX_CHARS = 32
MAX_SPRITES = 16
MAX_DRAPES = 8
MAX_THINGS = MAX_SPRITES + MAX_DRAPES
MAX_SCROLL_GROUPS = 4
ACTION_NONE = -1

OK, E_INVALID, E_UNSUPPORTED, E_HIP, E_STATE = 0, -1, -2, -3, -4

GAME_SCROLLY_MAZE, GAME_MARAUDERS, GAME_WAREHOUSE, GAME_HELLO_WORLD, GAME_WALKERS = 1, 2, 3, 4, 5
GAME_BETTER_SCROLLY = 6

PROG_NONE = 0
PROG_SM_PLAYER, PROG_SM_PATROLLER, PROG_SM_MAZE, PROG_SM_CASH = 10, 11, 12, 13
PROG_EM_PLAYER, PROG_EM_BUNKER, PROG_EM_MARAUDER, PROG_EM_UPBOLT, PROG_EM_DOWNBOLT = 20, 21, 22, 23, 24
PROG_WM_BOX, PROG_WM_JUDGE, PROG_WM_PLAYER = 30, 31, 32
PROG_HW_ROLLING, PROG_HW_SLIDING = 40, 41
PROG_WALKER, PROG_SCROLLY, PROG_STATIC = 50, 51, 52
PROG_BS_PLAYER, PROG_BS_PATROLLER, PROG_BS_CASH = 60, 61, 62
PROG_OD_PLAYER, PROG_OD_DRAGONDUCK, PROG_OD_SWORD = 70, 71, 72  # examples/ordeal.py
PLOT_WORDS = 4
PLOT_OD_HAS_SWORD, PLOT_OD_LAST_POSITION, PLOT_OD_PRIOR_CHAPTER = 0, 1, 2

CROP_FIXED, CROP_SCROLLING = 1, 2


class SpriteDesc(ctypes.Structure):
  _fields_ = [('ch', c_u8), ('is_walker', c_u8), ('visible', c_u8),
              ('prior_visible', c_u8), ('confined', c_u8), ('egocentric', c_u8),
              ('scrolling_group', c_u8), ('pad0', c_u8), ('program', c_i32),
              ('row', c_i32), ('col', c_i32), ('vrow', c_i32), ('vcol', c_i32),
              ('impassable', c_u8 * 16), ('param', c_i32 * 4)]


class DrapeDesc(ctypes.Structure):
  _fields_ = [('ch', c_u8), ('is_scrolly', c_u8), ('have_margins', c_u8),
              ('scrolling_group', c_u8), ('program', c_i32),
              ('curtain', c_u8_p), ('pattern', c_u8_p),
              ('pattern_rows', c_i32), ('pattern_cols', c_i32),
              ('corner_row', c_i32), ('corner_col', c_i32),
              ('margin_rows', c_i32), ('margin_cols', c_i32),
              ('param', c_i32 * 4)]


MAX_DIRECTIVES = 32
DIR_ADD_REWARD, DIR_TERMINATE, DIR_Z_ORDER, DIR_NEXT_CHAPTER = 1, 2, 3, 4
CHAPTER_NONE, CHAPTER_UNSET = -1, -2147483648


class Directive(ctypes.Structure):
  _fields_ = [('ch', c_u8), ('kind', c_u8), ('move_this', c_u8), ('in_front_of', c_u8),
              ('selector', c_i32), ('reward', c_i32), ('discount', ctypes.c_float)]


class Template(ctypes.Structure):
  _fields_ = [('abi_version', c_u32), ('game', c_i32),
              ('rows', c_i32), ('cols', c_i32),
              ('occlusion_in_layers', c_i32),
              ('n_chars', c_i32), ('chars', c_u8 * MAX_CHARS),
              ('backdrop', c_u8_p),
              ('n_sprites', c_i32), ('sprites', SpriteDesc * MAX_SPRITES),
              ('n_drapes', c_i32), ('drapes', DrapeDesc * MAX_DRAPES),
              ('n_things', c_i32),
              ('z_order', c_u8 * MAX_THINGS), ('schedule', c_u8 * MAX_THINGS),
              ('group_of', c_u8 * MAX_THINGS),
              ('n_groups', c_i32), ('n_actions', c_i32),
              ('param', c_i32 * 8),
              ('n_directives', c_i32), ('directives', Directive * MAX_DIRECTIVES),
              ('reward_is_float', c_i32), ('n_plot_words', c_i32)]


class Buffers(ctypes.Structure):
  _fields_ = [('batch', c_i64), ('rows', c_i32), ('cols', c_i32),
              ('n_chars', c_i32),
              ('planes', ctypes.c_void_p), ('reward', ctypes.c_void_p),
              ('reward_set', ctypes.c_void_p), ('discount', ctypes.c_void_p),
              ('done', ctypes.c_void_p), ('frame', ctypes.c_void_p),
              ('error', ctypes.c_void_p)]


class SpriteState(ctypes.Structure):
  _fields_ = [('row', c_i32), ('col', c_i32), ('vrow', c_i32), ('vcol', c_i32),
              ('visible', c_u8), ('pad', c_u8 * 3)]


class CropperDesc(ctypes.Structure):
  _fields_ = [('kind', c_i32), ('rows', c_i32), ('cols', c_i32),
              ('top', c_i32), ('left', c_i32), ('pad_char', c_i32),
              ('n_track', c_i32), ('to_track', c_u8 * MAX_THINGS),
              ('margin_rows', c_i32), ('margin_cols', c_i32),
              ('initial_offset_rows', c_i32), ('initial_offset_cols', c_i32),
              ('saccade', c_i32)]


POST_TO_ARRAY, POST_FEATURE_ARRAY, POST_REPAINT = 1, 2, 3
U8, I32, F32, I64, F64 = 1, 2, 3, 4, 5
POST_MAX_DEPTH = 32


class PlanesView(ctypes.Structure):
  _fields_ = [('planes', ctypes.c_void_p), ('batch', c_i64), ('rows', c_i32), ('cols', c_i32),
              ('pitch', c_i32), ('n_chars', c_i32), ('chars', c_u8 * MAX_CHARS)]


class PostDesc(ctypes.Structure):
  _fields_ = [('kind', c_i32), ('dtype', c_i32), ('depth', c_i32),
              ('lut', (c_u64 * 128) * POST_MAX_DEPTH), ('mapped', c_u8 * 128),
              ('chars', c_u8 * POST_MAX_DEPTH), ('stride', c_i64 * 3)]


class EpilogueDesc(ctypes.Structure):
  _fields_ = [('depth', c_i32), ('chars', c_u8 * POST_MAX_DEPTH), ('out_dev', ctypes.c_void_p), ('skip_layers', c_i32),
              ('channels_last', c_i32), ('to_array', c_i32), ('dtype', c_i32), ('lut', ctypes.c_void_p),
              ('mapped', ctypes.c_void_p)]


# Every symbol include/pcx.h declares: (name, restype, argtypes).
_VP = ctypes.c_void_p
SYMBOLS = [
    ('pcx_engine_create', c_i32, [ctypes.POINTER(Template), c_i64, c_i32, ctypes.POINTER(_VP)]),
    ('pcx_engine_destroy', None, [_VP]),
    ('pcx_engine_reset', c_i32, [_VP, _VP, _VP]),
    ('pcx_engine_step', c_i32, [_VP, _VP, c_i32, _VP]),
    ('pcx_engine_step_n', c_i32, [_VP, _VP, c_i32, c_i32, _VP]),
    ('pcx_engine_step_hashed', c_i32, [_VP, c_u64, c_i64, c_i64, c_i32, c_i32, _VP]),
    ('pcx_engine_buffers', c_i32, [_VP, ctypes.POINTER(Buffers)]),
    ('pcx_engine_bind_buffers', c_i32, [_VP, ctypes.POINTER(Buffers)]),
    ('pcx_engine_read_things', c_i32, [_VP, c_i64, c_i64, _VP, _VP]),
    ('pcx_engine_error_poll', c_i32, [_VP, _VP, ctypes.POINTER(c_i32)]),
    ('pcx_engine_errors_seen', c_i32, [_VP, _VP, c_i32]),
    ('pcx_engine_next_chapter', c_i32, [_VP, _VP]),
    ('pcx_engine_plot_words', c_i32, [_VP, _VP]),
    ('pcx_engine_set_plot_words', c_i32, [_VP, _VP, _VP]),
    ('pcx_engine_state_size', c_i32, [_VP, c_i32, ctypes.POINTER(c_u64)]),
    ('pcx_engine_export_state', c_i32, [_VP, _VP, c_u64, c_i32]),
    ('pcx_engine_import_state', c_i32, [_VP, _VP, c_u64]),
    ('pcx_engine_set_epilogue', c_i32, [_VP, ctypes.POINTER(EpilogueDesc)]),
    ('pcx_memcpy_d2h', c_i32, [_VP, _VP, c_u64]),
    ('pcx_memcpy_h2d', c_i32, [_VP, _VP, c_u64]),
    ('pcx_device_malloc', c_i32, [ctypes.POINTER(_VP), c_u64]),
    ('pcx_device_free', c_i32, [_VP]),
    ('pcx_stream_synchronize', c_i32, [_VP]),
    ('pcx_device_fill_probe', c_i32, [_VP, c_u64, _VP]),
    ('pcx_action_hash', c_u32, [c_u64, c_u64, c_u64]),
    ('pcx_engine_plane_pitch', c_i32, [_VP]),
    ('pcx_engine_bytes_per_step', c_i64, [_VP]),
    ('pcx_engine_kernel_name', ctypes.c_char_p, [_VP]),
    ('pcx_engine_launch_shape', c_i32, [_VP]),
    ('pcx_engine_tuner_done', c_i32, [_VP]),
    ('pcx_generic_specialise_check', c_i32, [ctypes.POINTER(Template), ctypes.c_char_p, c_i64, ctypes.POINTER(c_i64)]),
    ('pcx_scrolly_maze_specialise_check', c_i32, [ctypes.POINTER(Template), ctypes.c_char_p, c_i64, ctypes.POINTER(c_i64)]),
    ('pcx_engine_debug_counters', c_i32, [_VP, ctypes.POINTER(c_u32), c_i64]),
    ('pcx_debug_scrolly_consts', c_i64, [ctypes.POINTER(Template), c_i32, ctypes.POINTER(c_u32), c_i64]),
    ('pcx_last_error', ctypes.c_char_p, []),
    ('pcx_abi_version', c_u32, []),
    ('pcx_cropper_create', c_i32, [_VP, ctypes.POINTER(CropperDesc), ctypes.POINTER(_VP)]),
    ('pcx_cropper_destroy', None, [_VP]),
    ('pcx_cropper_crop', c_i32, [_VP, _VP]),
    ('pcx_cropper_buffers', c_i32, [_VP, ctypes.POINTER(_VP), ctypes.POINTER(_VP)]),
    ('pcx_cropper_errors', c_i32, [_VP, _VP]),
    ('pcx_cropper_plane_pitch', c_i32, [_VP]),
    ('pcx_cropper_set_features', c_i32, [_VP, _VP]),
    ('pcx_cropper_state_size', c_i32, [_VP, c_i32, ctypes.POINTER(c_u64)]),
    ('pcx_cropper_export_state', c_i32, [_VP, _VP, c_u64, c_i32]),
    ('pcx_cropper_import_state', c_i32, [_VP, _VP, c_u64]),
    ('pcx_cropper_bind_output', c_i32, [_VP, _VP]),
    ('pcx_engine_fuse_croppers', c_i32, [_VP, ctypes.POINTER(_VP), c_i32, c_i32, _VP]),
    ('pcx_cropper_error_buffer', c_i32, [_VP, ctypes.POINTER(_VP)]),
    ('pcx_cropper_error_poll', c_i32, [_VP, _VP, ctypes.POINTER(c_i32)]),
    ('pcx_engine_planes_view', c_i32, [_VP, ctypes.POINTER(PlanesView)]),
    ('pcx_cropper_planes_view', c_i32, [_VP, ctypes.POINTER(PlanesView)]),
    ('pcx_post_create', c_i32, [ctypes.POINTER(PlanesView), ctypes.POINTER(PostDesc), c_i32, ctypes.POINTER(_VP)]),
    ('pcx_post_destroy', None, [_VP]),
    ('pcx_post_run', c_i32, [_VP, _VP]),
    ('pcx_post_output', c_i32, [_VP, ctypes.POINTER(_VP), ctypes.POINTER(c_u64)]),
    ('pcx_post_errors', c_i32, [_VP, _VP]),
    ('pcx_post_plane_pitch', c_i32, [_VP]),
    ('pcx_post_bind_output', c_i32, [_VP, _VP, c_u64]),
    ('pcx_post_error_buffer', c_i32, [_VP, ctypes.POINTER(_VP)]),
    ('pcx_post_error_poll', c_i32, [_VP, _VP, ctypes.POINTER(c_i32)]),
    ('pcx_gather_create', c_i32, [ctypes.POINTER(_VP), c_i32, ctypes.POINTER(_VP)]),
    ('pcx_gather_destroy', None, [_VP]),
    ('pcx_gather_scalars', c_i32, [_VP, ctypes.POINTER(_VP)]),
    ('pcx_gather_buffers', c_i32, [_VP, c_i32, ctypes.POINTER(_VP), ctypes.POINTER(c_i64)]),
]

# PCX_LIB selects another build of the same library (A/B kernel experiments).
LIB_PATH = os.environ.get('PCX_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'libpcx.so')


class NativeLibraryMissing(RuntimeError):
  pass


def bind(lib, symbols, rename=None):
  """Attach restype/argtypes for every (name, restype, argtypes) entry."""
  for name, restype, argtypes in symbols:
    real = rename(name) if rename else name
    if os.environ.get('PCX_LIB') and not hasattr(lib, real):
      continue  # an older build selected for an A/B run: calling what it lacks raises AttributeError then
    fn = getattr(lib, real)  # AttributeError if the symbol is missing
    fn.restype = restype
    fn.argtypes = argtypes
  return lib


_lib = None


def lib():
  """The HIP engine library; raises loudly when it has not been built."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise NativeLibraryMissing(
          '{} is missing: build it with `python __graft_entry__.py build` (or '
          '`make -C pycolab_amd/csrc`). pycolab_amd has no CPU fallback.'.format(
              LIB_PATH))
    from pycolab_amd import _torchprobe
    _torchprobe.torch_module()  # settle which HIP runtime the process uses first
    _lib = bind(ctypes.CDLL(LIB_PATH), SYMBOLS)
    if _lib.pcx_abi_version() != ABI_VERSION:
      raise NativeLibraryMissing('libpcx.so ABI version mismatch; rebuild it')
  return _lib


class PcxError(RuntimeError):
  pass


def check(code):
  if code != 0:
    message = lib().pcx_last_error().decode('utf-8', 'replace')
    if code == E_UNSUPPORTED:
      raise NotImplementedError(message)
    if code == E_INVALID:
      raise ValueError(message)
    raise PcxError('pcx error {}: {}'.format(code, message))
