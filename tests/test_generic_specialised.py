"""pcx_generic_step built per template at run time (pcx_generic.hip rtc::, pcx_generic_kernel.h PCX_GENERIC_SPEC): the
template's tables, schedule and z-order as compile-time constants, compiled by hiprtc when a large engine is created.

CPU leg: the embedded device sources compile for the golden templates (hiprtc needs no device), the code object is
cached on disk and found again, templates the table-driven kernel refuses are refused here too.
GPU leg: the specialised build steps every fixture exactly as the oracle does (reference: engine.py:583-847 around
the fixtures' Sprites and Drapes, via oracle/pcx_oracle.c) and exactly as the table-driven build does, with fused
croppers, plot directives, scrolling groups, z-order changes, resets left to the caller and action tapes."""
import ctypes
import os
import time

import numpy as np
import pytest

from pycolab_amd import _native as N
from tests import helpers

GENERIC = ['warehouse_L0', 'warehouse_L2', 'warehouse_custom_B', 'warehouse_L0_unoccluded', 'marauders', 'marauders_custom_A',
           'marauders_unoccluded', 'walkers_room', 'walkers_hidden', 'walkers_scroll_groups', 'walkers_scroll_always', 'walkers_scroll_margins',
           'directives_z_order', 'directives_reward_discount', 'hello_world', 'hello_custom_A', 'better_scrolly_maze_L1',
           'better_scrolly_custom_B',
           # round 6: a float32 reward lane (tabled entities adding Python floats; the three chapters of examples/ordeal.py: tests/test_ordeal.py)
           'directives_float_rewards', 'ordeal_castle']


def check(template, cache):
  os.environ['PCX_JIT_CACHE'] = str(cache)
  try:
    ct, keep = template.to_ctypes()
    log = ctypes.create_string_buffer(8192)
    n = ctypes.c_int64(0)
    t0 = time.time()
    rc = N.lib().pcx_generic_specialise_check(ctypes.byref(ct), log, len(log), ctypes.byref(n))
    return rc, int(n.value), log.value.decode(), time.time() - t0
  finally:
    del os.environ['PCX_JIT_CACHE']


@pytest.mark.parametrize('name', ['warehouse_L0', 'marauders_custom_A', 'walkers_scroll_groups', 'directives_z_order', 'hello_world',
                                  'marauders_unoccluded'])
def test_specialised_build_compiles_without_a_device(name, tmp_path):
  rc, size, log, _ = check(helpers.load_template(name), tmp_path)
  assert rc == 0, log[:2000]
  assert size > 4096
  files = [f for f in os.listdir(tmp_path) if f.endswith('.hsaco')]
  assert len(files) == 1 and os.path.getsize(tmp_path / files[0]) == size


def test_every_golden_template_the_kernel_accepts_has_a_specialised_build(tmp_path):
  """All of tests/golden/templates: a template either is refused by the table-driven kernel's plan (the scrolly_maze
  programs have no device program there) or compiles -- no template may plan and then fail to build."""
  import glob
  from concurrent.futures import ThreadPoolExecutor
  built = refused = 0
  names = [os.path.basename(path)[:-4] for path in sorted(glob.glob(os.path.join(helpers.ROOT, 'tests', 'golden', 'templates', '*.npz')))]
  os.environ['PCX_JIT_CACHE'] = str(tmp_path)  # (check() sets and clears it around every call: here four threads compile side by side)

  def one(name):
    ct, keep = helpers.load_template(name).to_ctypes()
    log, n = ctypes.create_string_buffer(8192), ctypes.c_int64(0)
    rc = N.lib().pcx_generic_specialise_check(ctypes.byref(ct), log, len(log), ctypes.byref(n))  # (ctypes releases the GIL; hiprtc compiles independent programs)
    return rc, int(n.value), log.value.decode()

  try:
    with ThreadPoolExecutor(4) as pool:
      results = list(pool.map(one, names))
  finally:
    del os.environ['PCX_JIT_CACHE']
  for name, (rc, size, log) in zip(names, results):
    if rc == 0:
      assert size > 4096, name
      built += 1
    else:
      assert rc == N.E_UNSUPPORTED and 'hiprtc' not in log and name.startswith('scrolly_'), (name, log[:500])
      refused += 1
  assert built >= 24 and refused >= 10, (built, refused)
  # The build's assembly scan (tools/sgpr_hazard_scan.py: an inline-asm VMEM instruction reading an SGPR a VALU wrote
  # less than five wait states earlier) cannot see these kernels -- they are compiled at run time -- so the code
  # objects are disassembled and scanned here.
  import importlib.util
  spec = importlib.util.spec_from_file_location('sgpr_hazard_scan', os.path.join(helpers.ROOT, 'tools', 'sgpr_hazard_scan.py'))
  scan = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(scan)
  if not os.path.exists(scan.OBJDUMP):
    pytest.skip('no llvm-objdump at ' + scan.OBJDUMP)
  objects = sorted(f for f in os.listdir(tmp_path) if f.endswith('.hsaco'))
  assert len(objects) == built
  for f in objects:
    lines = scan.disassemble(str(tmp_path / f))
    assert sum('global_load_lds_dword' in l for l in lines) > 10 and any(l.startswith('_Zpcx_generic_step') for l in lines), f
    body = [(i, l) for i, l in enumerate(lines, 1)]
    assert scan.scan_kernel(f, 'pcx_generic_step', body) == [], f


def test_code_objects_are_cached_by_content(tmp_path):
  t = helpers.load_template('walkers_room')
  rc, size, _, cold = check(t, tmp_path)
  assert rc == 0
  rc, again, _, warm = check(t, tmp_path)
  assert rc == 0 and again == size and warm < cold  # (a file read against a compile)
  # the seed and the environment offset are per engine, not per template: same code object
  t.param[0], t.param[2] = 12345, 999
  assert check(t, tmp_path)[:2] == (0, size)
  assert len(os.listdir(tmp_path)) == 1
  # another template, another entry
  assert check(helpers.load_template('warehouse_custom_A'), tmp_path)[0] == 0
  assert len(os.listdir(tmp_path)) == 2


def test_prebuild_from_python(tmp_path, monkeypatch):
  """GameTemplate.prebuild(): a deployment fills the cache without a GPU (and a game built from pycolab's own API does)."""
  monkeypatch.setenv('PCX_JIT_CACHE', str(tmp_path))
  size = helpers.load_template('directives_z_order').prebuild()
  assert size > 4096 and [os.path.getsize(tmp_path / f) for f in os.listdir(tmp_path)] == [size]
  with pytest.raises(NotImplementedError):
    helpers.load_template('scrolly_maze_L1').prebuild()
  from oracle import directive_scenarios as ds
  from pycolab_amd import ascii_art
  from pycolab_amd.prefab_parts import tabled
  game = ds.build_twin(ds.STORY[0], ascii_art, tabled)  # (a game assembled through ascii_art / the prefab parts)
  assert game.template.prebuild() > 4096
  assert len(os.listdir(tmp_path)) == 2


def test_templates_the_table_driven_kernel_refuses(tmp_path):
  rc, size, _, _ = check(helpers.load_template('scrolly_maze_L0'), tmp_path)  # (pcx_scrolly_maze_step's programs: no device program here)
  assert rc == N.E_UNSUPPORTED
  assert size == 0 and not os.listdir(tmp_path)


# ---- GPU leg -----------------------------------------------------------------------------------------------------------

def engines(name, batch, monkeypatch, tmp_path, **kw):
  from oracle import binding
  from tests.hip_adapter import HipAdapter
  monkeypatch.setenv('PCX_FORCE_GENERIC', '1')
  monkeypatch.setenv('PCX_JIT_CACHE', str(tmp_path))
  t = helpers.load_template(name)
  t.param[0] = 0xD1CE
  # (the native engine -- and with it the choice of build -- is made by its_showtime())
  monkeypatch.setenv('PCX_GENERIC_JIT', '1')
  spec = HipAdapter(t, batch, **kw)
  spec.reset()
  monkeypatch.setenv('PCX_GENERIC_JIT', '0')
  table = HipAdapter(t, batch, **kw)
  table.reset()
  monkeypatch.delenv('PCX_GENERIC_JIT')
  orc = binding.OracleEngine(t, batch)
  orc.reset()
  return t, spec, table, orc


def shape_of(hip):
  return int(N.lib().pcx_engine_launch_shape(hip.eng._native))


def same(hip, orc, where):
  for name in ('planes', 'reward', 'reward_set', 'discount', 'done', 'frame', 'error'):
    np.testing.assert_array_equal(hip.read(name), np.array(getattr(orc, name)), err_msg='%s: %s' % (where, name))
  np.testing.assert_array_equal(hip.sprites(), orc.sprites(), err_msg=where + ': sprites')
  np.testing.assert_array_equal(hip.curtains(), orc.curtains(), err_msg=where + ': curtains')


@pytest.mark.gpu
@pytest.mark.parametrize('name', GENERIC)
def test_specialised_build_matches_oracle_and_table_driven_build(name, monkeypatch, tmp_path):
  B, T = 64 * 9 + 21, 128
  t, spec, table, orc = engines(name, B, monkeypatch, tmp_path)
  assert shape_of(spec) == 31 and shape_of(table) == 30
  assert N.lib().pcx_engine_kernel_name(spec.eng._native).decode() == 'pcx_generic_step'
  same(spec, orc, 'frame 0')
  for t0 in range(0, T, 16):
    n = 1 if t0 < 16 else 16
    for u in range(t0, t0 + 16, n):
      spec.step_hashed(0x5EED, u, n); table.step_hashed(0x5EED, u, n); orc.step_hashed(0x5EED, u, n)
      same(spec, orc, 'specialised, after step %d' % (u + n))
    same(table, orc, 'table-driven, after step %d' % (t0 + 16))


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['warehouse_L0', 'marauders_custom_A', 'walkers_scroll_groups', 'directives_z_order', 'directives_float_rewards', 'ordeal_castle'])
def test_specialised_build_tapes_and_environments_left_alone(name, monkeypatch, tmp_path):
  B, T = 500, 90
  t, spec, table, orc = engines(name, B, monkeypatch, tmp_path)
  rng = np.random.RandomState(3)
  na = max(1, int(t.n_actions))
  for step in range(T):
    a = rng.randint(0, na, size=B).astype(np.int32)
    r = rng.rand(B)
    a[r < 0.05] = -1
    a[(r >= 0.05) & (r < 0.08)] = na + rng.randint(0, 30)
    auto = step % 4 != 0
    spec.step(a, auto_reset=auto); orc.step(a, auto_reset=auto)
    same(spec, orc, 'step %d' % step)
  assert shape_of(spec) == 31


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['warehouse_L0', 'marauders_custom_A', 'walkers_scroll_groups'])
def test_specialised_build_at_timing_size_against_table_driven_build(name, monkeypatch, tmp_path):
  """32,768 environments (what tools/generic_timing.py times and the default policy specialises): every byte equal to
  the table-driven build's, the first and the last 1,024 environments equal to the oracle's."""
  import torch
  from oracle import binding
  B, T, K = 32768, 48, 1024
  t, spec, table, _ = engines(name, B, monkeypatch, tmp_path)
  spec.step_hashed(0xFACE, 0, T); table.step_hashed(0xFACE, 0, T)
  assert shape_of(spec) == 31 and shape_of(table) == 30
  assert torch.equal(spec.eng.planes_view(), table.eng.planes_view())
  for nm in ('reward', 'reward_set', 'discount', 'done', 'frame', 'error'):
    assert torch.equal(spec.eng.buffers[nm].tensor, table.eng.buffers[nm].tensor), nm
  tt = helpers.load_template(name)  # (the RNG of a game that draws is keyed by the GLOBAL environment index: param[2..3])
  tt.param[0], tt.param[2] = t.param[0], B - K
  head, tail = binding.OracleEngine(t, K), binding.OracleEngine(tt, K)
  head.reset(); tail.reset()
  head.step_hashed(0xFACE, 0, T); tail.step_hashed(0xFACE, 0, T, env_offset=B - K)
  planes = spec.eng.planes_view()
  np.testing.assert_array_equal(planes[:K].cpu().numpy(), np.array(head.planes))
  np.testing.assert_array_equal(planes[B - K:].cpu().numpy(), np.array(tail.planes))
  for nm in ('reward', 'done', 'frame'):
    got = spec.eng.buffers[nm].tensor
    np.testing.assert_array_equal(got[:K].cpu().numpy(), np.array(getattr(head, nm)))
    np.testing.assert_array_equal(got[B - K:].cpu().numpy(), np.array(getattr(tail, nm)))


@pytest.mark.gpu
def test_default_policy_specialises_large_engines_only(monkeypatch, tmp_path):
  from tests.hip_adapter import HipAdapter
  monkeypatch.setenv('PCX_FORCE_GENERIC', '1')
  monkeypatch.setenv('PCX_JIT_CACHE', str(tmp_path))
  monkeypatch.delenv('PCX_GENERIC_JIT', raising=False)
  t = helpers.load_template('walkers_room')
  small, large = HipAdapter(t, 512), HipAdapter(t, 4096)
  small.reset(); large.reset()
  assert shape_of(small) == 30 and shape_of(large) == 31
  monkeypatch.setenv('PCX_GENERIC_JIT_MIN', '100000')
  other = HipAdapter(t, 4096)
  other.reset()
  assert shape_of(other) == 30


@pytest.mark.gpu
def test_a_damaged_cache_entry_is_recompiled(monkeypatch, tmp_path):
  """A code object that will not load (a file cut short on disk) is compiled again and replaced: the engine still gets
  its specialised build."""
  from tests.hip_adapter import HipAdapter
  monkeypatch.setenv('PCX_FORCE_GENERIC', '1')
  monkeypatch.setenv('PCX_GENERIC_JIT', '1')
  t = helpers.load_template('walkers_room')
  # (a board of its own: a build some earlier test of this process loaded would be served from memory, not from the cache)
  free = np.argwhere(t.backdrop == ord(' '))
  t.backdrop[tuple(free[len(free) // 2])] = ord('w')
  assert check(t, tmp_path)[0] == 0
  monkeypatch.setenv('PCX_JIT_CACHE', str(tmp_path))
  (entry,) = [f for f in os.listdir(tmp_path) if f.endswith('.hsaco')]
  good = os.path.getsize(tmp_path / entry)
  with open(tmp_path / entry, 'wb') as f:
    f.write(b'not a code object')
  hip = HipAdapter(t, 300)
  hip.reset()
  assert shape_of(hip) == 31
  assert os.path.getsize(tmp_path / entry) == good
