"""pcx_scrolly_maze_step built per level at run time (round 6: pcx_scrolly_maze.hip jit::, pcx_scrolly_maze_kernel.h
PCX_SM_SPEC): a level of one's own on the example's 10x30 board with its 'abcP' cast -- what a new entry of
examples/scrolly_maze.py MAZES_ART is to the library (scrolly_maze.py:212-242) -- gets the two instances the shipped
levels have inside libpcx.so: persistent workers and the cooperative small-batch shape with the level's constants
compiled in.  The fixture is tests/golden/templates/scrolly_custom_H (oracle/custom_levels.py; its trace is recorded from
the reference like the others).

CPU leg: the embedded device sources compile for that level (hiprtc needs no device) into a code object that holds both
instances under the names the loader looks up, cached by content; the header of constants is what
pcx_debug_scrolly_consts answers (the function pcx_sm_shipped.h is generated from); shipped levels and other shapes are
refused with an empty log; the SGPR-hazard scan of the build runs over the code object.
GPU leg: the run-time instances step the level exactly as the oracle does (reference: engine.py:583-847 around the
example's Sprites and Drapes, via oracle/pcx_oracle.c) -- several worker shapes, single steps and launches of several,
ragged batches, the cooperative shape -- and exactly as the run-time-constants instances do over a large batch."""
import ctypes
import os
import re
import time

import numpy as np
import pytest

from pycolab_amd import _native as N
from tests import helpers

LEVEL = 'scrolly_custom_H'


def check(template, cache, **env):
  saved = {k: os.environ.get(k) for k in list(env) + ['PCX_JIT_CACHE']}
  os.environ['PCX_JIT_CACHE'] = str(cache)
  os.environ.update({k: str(v) for k, v in env.items()})
  try:
    ct, keep = template.to_ctypes()
    log = ctypes.create_string_buffer(8192)
    n = ctypes.c_int64(0)
    t0 = time.time()
    rc = N.lib().pcx_scrolly_maze_specialise_check(ctypes.byref(ct), log, len(log), ctypes.byref(n))
    return rc, int(n.value), log.value.decode(), time.time() - t0
  finally:
    for k, v in saved.items():
      if v is None:
        os.environ.pop(k, None)
      else:
        os.environ[k] = v


def planned_words(template, unit):
  ct, keep = template.to_ctypes()
  n = N.lib().pcx_debug_scrolly_consts(ctypes.byref(ct), unit, None, 0)
  assert n > 0
  words = (ctypes.c_uint32 * n)()
  assert N.lib().pcx_debug_scrolly_consts(ctypes.byref(ct), unit, words, n) == n
  return list(words)


def test_run_time_build_compiles_without_a_device_and_is_cached(tmp_path):
  t = helpers.load_template(LEVEL)
  dump = tmp_path / 'spec.h'
  cache = tmp_path / 'cache'
  rc, size, log, cold = check(t, cache, PCX_SM_DUMP_SPEC=dump)
  assert rc == 0, log[:2000]
  assert size > 16384
  files = [f for f in os.listdir(cache) if f.endswith('.hsaco')]
  assert len(files) == 1 and files[0].startswith('pcx_scrolly_maze_') and os.path.getsize(cache / files[0]) == size
  # the header of constants: the very words pcx_debug_scrolly_consts answers (tools/gen_sm_shipped.py writes pcx_sm_shipped.h
  # from the same entry), as the persistent shape launches them and as init() leaves them
  text = open(dump).read()
  arrays = {m.group(1): [int(w, 16) for w in re.findall(r'0x([0-9A-F]{8})u', m.group(2))]
            for m in re.finditer(r'#define (PCX_SM_SPEC\w*WORDS) ((?:.*\\\n)*.*)\n', text)}
  assert arrays['PCX_SM_SPEC_WORDS'] == planned_words(t, 64)
  assert arrays['PCX_SM_SPEC_PLAIN_WORDS'] == planned_words(t, 0)
  assert '#define PCX_SM_SPEC_N %d\n' % len(arrays['PCX_SM_SPEC_WORDS']) in text and 'PCX_SM_SPEC_NO_PS' not in text
  # found again: a file read against a compile
  rc, again, _, warm = check(t, cache)
  assert rc == 0 and again == size and warm < cold and len(os.listdir(cache)) == 1
  # the build's assembly scan cannot see a kernel compiled at run time: the code object is disassembled and scanned here
  # (tools/sgpr_hazard_scan.py: an inline-asm VMEM instruction reading an SGPR a VALU wrote less than five wait states before)
  import importlib.util
  spec = importlib.util.spec_from_file_location('sgpr_hazard_scan', os.path.join(helpers.ROOT, 'tools', 'sgpr_hazard_scan.py'))
  scan = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(scan)
  if not os.path.exists(scan.OBJDUMP):
    pytest.skip('no llvm-objdump at ' + scan.OBJDUMP)
  lines = scan.disassemble(str(cache / files[0]))
  kernels = [l for l in lines if l.startswith('_Z') and 'pcx_scrolly_maze_step' in l]
  assert len(kernels) == 2 and sum('global_load_lds_dword' in l for l in lines) > 10, kernels
  assert scan.scan_kernel(files[0], 'pcx_scrolly_maze_step', [(i, l) for i, l in enumerate(lines, 1)]) == []


def test_shipped_levels_and_other_games_are_refused_without_a_compile(tmp_path):
  # (unoccluded layers: the raw-mask render path exists with a run-time shape only -- such levels keep the shape-generic instance)
  for name in ('scrolly_maze_L0', 'scrolly_maze_L2', 'warehouse_L0', 'better_scrolly_maze_L1', 'scrolly_custom_A_unoccluded', 'scrolly_maze_L1_unoccluded'):
    rc, size, log, dt = check(helpers.load_template(name), tmp_path)
    assert rc == N.E_UNSUPPORTED and size == 0 and log == '' and dt < 5.0, (name, rc, log)
  # a board whose planes are not whole dwords (3 x 7 cells): the static-shape code paths do not take it
  t = helpers.load_template('scrolly_custom_D')
  t.rows, t.cols = 3, 7
  rc, size, log, _ = check(t, tmp_path)
  assert rc != 0 and size == 0
  assert not os.path.exists(tmp_path) or not os.listdir(tmp_path)
  with pytest.raises(NotImplementedError):
    helpers.load_template('scrolly_maze_L1').prebuild()


OTHER_SHAPES = ['scrolly_custom_A', 'scrolly_custom_B', 'scrolly_custom_C', 'scrolly_custom_D', 'scrolly_custom_E', 'scrolly_custom_F', 'scrolly_custom_G']


def test_levels_of_other_shapes_get_one_instance_with_the_shape_as_template_arguments(tmp_path):
  """Other boards, one to six sprites, other z-orders (oracle/custom_levels.py): libpcx.so steps them with
  the shape-generic instances (every stride a run-time value: 36-60 k instructions for four to six sprites); the run-time
  build is ONE instance with the shape as template arguments and the constants compiled in (launch shape 21)."""
  from concurrent.futures import ThreadPoolExecutor
  os.environ['PCX_JIT_CACHE'] = str(tmp_path)
  try:
    def one(name):
      ct, keep = helpers.load_template(name).to_ctypes()
      log, n = ctypes.create_string_buffer(8192), ctypes.c_int64(0)
      return N.lib().pcx_scrolly_maze_specialise_check(ctypes.byref(ct), log, len(log), ctypes.byref(n)), int(n.value), log.value.decode()
    with ThreadPoolExecutor(4) as pool:
      results = list(pool.map(one, OTHER_SHAPES))
  finally:
    del os.environ['PCX_JIT_CACHE']
  for name, (rc, size, log) in zip(OTHER_SHAPES, results):
    assert rc == 0 and 8192 < size < 131072, (name, rc, log[:1500])  # (a third of the shape-generic instance's code)
  assert len(os.listdir(tmp_path)) == len(OTHER_SHAPES)


def test_a_level_with_six_coin_words_builds_the_cooperative_instance_only(tmp_path):
  """The persistent shape holds at most four coin words in a worker's inbox (ScrollyMazeBackend::ps_shape): the shipped level
  2 -- taken as a level of one's own (PCX_SM_BAKED=0: no instance of the library may claim it) -- gets the cooperative
  instance alone."""
  t = helpers.load_template('scrolly_maze_L2')
  dump = tmp_path / 'spec.h'
  rc, size, log, _ = check(t, tmp_path / 'cache', PCX_SM_BAKED=0, PCX_SM_DUMP_SPEC=dump)
  assert rc == 0 and size > 8192, log[:2000]
  assert '#define PCX_SM_SPEC_NO_PS 1' in open(dump).read()


def test_prebuild_from_python_covers_scrolly_maze_levels_of_ones_own(tmp_path, monkeypatch):
  monkeypatch.setenv('PCX_JIT_CACHE', str(tmp_path))
  size = helpers.load_template(LEVEL).prebuild()
  assert size > 16384 and [os.path.getsize(tmp_path / f) for f in os.listdir(tmp_path)] == [size]


# ---- GPU ------------------------------------------------------------------------------------------------------------------


def _gpu():
  from oracle import binding
  from tests.hip_adapter import HipAdapter
  from tests.test_persistent_shapes import Knobs, OracleAdapter, assert_same, raw_shape_of
  return binding, HipAdapter, Knobs, OracleAdapter, assert_same, raw_shape_of


@pytest.mark.gpu
@pytest.mark.parametrize('jit,waves,lock,grid,dynamic', [(1, 2, 1, 5, 1), (0, 2, 1, 5, 1), (1, 3, 2, 3, 0), (1, 6, 2, 1, 1), (1, 4, 2, 4096, 1)])
def test_persistent_workers_on_the_run_time_instance_match_oracle(jit, waves, lock, grid, dynamic):
  """Launch shape 7 (PCX_SM_JIT=1 whatever the batch; 0: the run-time-constants instance, shape 3) against the oracle on a
  ragged batch: single steps, then launches of eight (13), resets included."""
  _, HipAdapter, Knobs, OracleAdapter, assert_same, raw_shape_of = _gpu()
  t = helpers.load_template(LEVEL)
  B, T = 2999, 160
  with Knobs(PCX_COOP_BELOW=0, PCX_SM_SHAPE=3, PCX_SM_JIT=jit, PCX_SM_WAVES=waves, PCX_SM_LOCK=lock, PCX_SM_GRID=grid, PCX_SM_DYNAMIC=dynamic):
    hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
    hip.reset(); orc.reset()
    assert_same(hip, orc, 'frame 0')
    t0 = 0
    while t0 < T:
      n = 1 if t0 < 24 else 8
      hip.step_hashed(0x5EED, t0, n); orc.step_hashed(0x5EED, t0, n)
      assert raw_shape_of(hip) == (13 if n > 1 else 7 if jit else 3)
      t0 += n
      assert_same(hip, orc, 'jit %d after step %d' % (jit, t0))
    assert int(orc.read('frame').min()) < T


@pytest.mark.gpu
@pytest.mark.parametrize('jit', [1, 0])
@pytest.mark.parametrize('B', [250, 4096, 16391])
def test_cooperative_shape_on_the_run_time_instance_matches_oracle(B, jit):
  """Small batches: the cooperative instance of the run-time build (16 / 32 / 64 environments per workgroup, single launches
  and launches of several steps) against the oracle; PCX_SM_JIT=0: the instance that reads the constants from its arguments."""
  _, HipAdapter, Knobs, OracleAdapter, assert_same, raw_shape_of = _gpu()
  t = helpers.load_template(LEVEL)
  with Knobs(PCX_SM_JIT=jit):
    hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
    hip.reset(); orc.reset()
    t0 = 0
    for n in ((1, 1, 1, 7, 64, 3, 1, 60) if B < 10000 else (1, 1, 7, 33, 1)):  # (the largest batch: fewer steps, the oracle is what it waits for)
      hip.step_hashed(0x5EED, t0, n); orc.step_hashed(0x5EED, t0, n)
      if os.environ.get('PCX_COOP_BELOW') != '0':  # (a suite run with the cooperative shape forced off compares results only)
        assert raw_shape_of(hip) == (12 if n > 1 else 10)
      t0 += n
      assert_same(hip, orc, 'B %d jit %d after step %d' % (B, jit, t0))


@pytest.mark.gpu
def test_a_large_batch_takes_the_run_time_instance_by_itself_and_equals_the_other_instance_everywhere():
  """131,072 environments, default knobs: the engine compiles (or finds) its level's instances at creation and its steps
  take launch shape 7; what they write equals, over the WHOLE batch, what the run-time-constants instance writes for the
  same tape (PCX_SM_JIT=0: shape 3); the first and the last 1,024 environments against the oracle."""
  import torch
  _, HipAdapter, Knobs, OracleAdapter, assert_same, raw_shape_of = _gpu()
  t = helpers.load_template(LEVEL)
  B, T, K = 131072, 24, 1024
  hip = HipAdapter(t, B)
  hip.reset()
  for t0 in range(T // 2):
    hip.step_hashed(0xC0FFEE, t0, 1)
    assert raw_shape_of(hip) == 7
  hip.step_hashed(0xC0FFEE, T // 2, T - T // 2)
  assert raw_shape_of(hip) == 13
  with Knobs(PCX_SM_JIT=0):
    ref = HipAdapter(t, B)
    ref.reset()
    for t0 in range(T // 2):
      ref.step_hashed(0xC0FFEE, t0, 1)
      assert raw_shape_of(ref) == 3
    ref.step_hashed(0xC0FFEE, T // 2, T - T // 2)
  assert torch.equal(hip.eng.planes_view(), ref.eng.planes_view())
  for name in ('reward', 'reward_set', 'discount', 'done', 'frame', 'error'):
    assert torch.equal(hip.eng.buffers[name].tensor, ref.eng.buffers[name].tensor), name
  del ref
  planes = hip.eng.planes_view()
  for off in (0, B - K):
    orc = OracleAdapter(t, K)
    orc.reset()
    orc.step_hashed(0xC0FFEE, 0, T, env_offset=off)
    np.testing.assert_array_equal(planes[off:off + K].cpu().numpy(), orc.read('planes'), err_msg='environments from %d' % off)
    for name in ('reward', 'reward_set', 'discount', 'done', 'frame'):
      np.testing.assert_array_equal(hip.eng.buffers[name].tensor[off:off + K].cpu().numpy(), orc.read(name), err_msg=name)
  assert not hip.eng.buffers['error'].tensor.any()


@pytest.mark.gpu
@pytest.mark.parametrize('name', OTHER_SHAPES)
def test_run_time_instances_of_other_shapes_match_the_reference_and_the_oracle(name, monkeypatch):
  """The reference's trace of the level and 96 steps of 1,100 environments against the oracle, single steps and launches left to
  the engine, through the instance compiled for the level (PCX_SM_JIT=1 whatever the batch): launch shape 21."""
  _, HipAdapter, Knobs, OracleAdapter, assert_same, raw_shape_of = _gpu()
  monkeypatch.setenv('PCX_SM_JIT', '1')
  helpers.replay_trace(HipAdapter, helpers.load_trace(name))
  t = helpers.load_template(name)
  B = 1100
  hip, orc = HipAdapter(t, B), OracleAdapter(t, B)
  hip.reset(); orc.reset()
  assert_same(hip, orc, 'frame 0')
  t0 = 0
  while t0 < 96:
    n = 1 if t0 < 24 else 8
    hip.step_hashed(0x5EED, t0, n); orc.step_hashed(0x5EED, t0, n)
    assert raw_shape_of(hip) == 21
    t0 += n
    assert_same(hip, orc, '%s after step %d' % (name, t0))
