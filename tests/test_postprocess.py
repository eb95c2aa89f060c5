"""Observation post-processors (rendering.py:304-661): the numpy oracle (CPU)
and the device kernels (GPU) against outputs of the reference's own classes
recorded in the golden traces."""
import json

import numpy as np
import pytest

from oracle import postprocess as opost
from pycolab_amd import rendering
from tests import helpers

POSTED = ['marauders', 'warehouse_L1']
# games only the table-driven kernel steps: the reference's feature arrays (and one value array) recorded on them (round 4)
POSTED_TABLE_DRIVEN = ['walkers_scroll_groups', 'directives_z_order', 'marauders_custom_A']


def specs_of(trace):
  return json.loads(bytes(trace['post_specs']).decode()), int(trace['post_every'][0])


def frames_of(T, every):
  return [0] + [t for t in range(1, T + 1) if t % every == 0]


def make_post(spec):
  if spec['kind'] == 'repaint':
    return rendering.ObservationCharacterRepainter(dict(spec['mapping']))
  if spec['kind'] == 'features':
    return rendering.ObservationToFeatureArray(list(spec['layers']), permute=spec['permute'])
  return rendering.ObservationToArray(dict(spec['mapping']), dtype=np.dtype(spec['dtype']), permute=spec['permute'])


@pytest.mark.parametrize('name', POSTED + POSTED_TABLE_DRIVEN)
def test_oracle_postprocessors_match_reference(name):
  tr = helpers.load_trace(name)
  specs, every = specs_of(tr)
  chars = [chr(c) for c in tr['chars']]
  T, E = tr['actions'].shape
  for fi, t in enumerate(frames_of(T, every)):
    for e in range(0, E, 5):
      board = tr['boards'][t, e]
      layers = {c: board == ord(c) for c in chars}
      for i, sp in enumerate(specs):
        want = tr['post_%d' % i][fi, e]
        if sp['kind'] == 'repaint':
          got, _ = opost.repaint(board, chars, sp['mapping'])
        elif sp['kind'] == 'features':
          got = opost.feature_array(layers, list(sp['layers']), board.shape, sp['permute'])
        else:
          got = opost.to_array(board, sp['mapping'], sp['dtype'], sp['permute'])
        assert got.dtype == want.dtype and got.shape == want.shape
        np.testing.assert_array_equal(got, want)


def test_fuse_into_answers_false_without_a_device_engine():
  """fuse_into() never raises for "cannot": an engine that is not in play on a device (no native handle yet), a
  permuted axis order other than channels last.  (The device-side refusals are in the GPU tests.)"""
  from pycolab_amd.engine import Engine
  eng = Engine.from_template(helpers.load_template('scrolly_maze_L0'), batch=4)
  assert eng._native is None
  assert rendering.ObservationToArray({'a': 1.0}, dtype=np.float32).fuse_into(eng) is False
  assert rendering.ObservationToArray({'a': (1, 2, 3)}, dtype=np.uint8, permute=(2, 0, 1)).fuse_into(eng) is False
  assert rendering.ObservationToFeatureArray('P').fuse_into(eng) is False
  assert rendering.ObservationToFeatureArray('P', permute=(2, 0, 1)).fuse_into(eng) is False
  assert rendering.ObservationCharacterRepainter({'a': 'b'}).fuse_into(eng) is False
  assert eng._epilogue is None and not eng._epilogue_only


def test_constructor_guards():
  with pytest.raises(ValueError):
    rendering.ObservationToArray({'a': [1, 2]}, permute=(0, 1))
  with pytest.raises(ValueError):
    rendering.ObservationToArray({'a': 1}, permute=(0, 1, 2))
  with pytest.raises(ValueError):
    rendering.ObservationToFeatureArray('ab', permute=(0, 1))
  with pytest.raises(TypeError):   # needs an observation that knows its planes
    rendering.ObservationToFeatureArray('ab')(rendering.Observation(board=np.zeros((2, 2)), layers={'a': 0}))


@pytest.mark.gpu
@pytest.mark.parametrize('name', POSTED + POSTED_TABLE_DRIVEN)
def test_device_postprocessors_match_reference(name):
  from pycolab_amd.engine import Engine
  tr = helpers.load_trace(name)
  specs, every = specs_of(tr)
  t = helpers.load_template(tr['template'])
  T, E = tr['actions'].shape
  eng = Engine.from_template(t, batch=E, auto_reset=True, seed=helpers.GOLDEN_RNG_SEED)
  posts = [make_post(sp) for sp in specs]
  obs = eng.its_showtime()[0]
  fi = 0
  for step in range(T + 1):
    if step:
      obs = eng.play(tr['actions'][step - 1])[0]
    if step == 0 or step % every == 0:
      for i, (sp, po) in enumerate(zip(specs, posts)):
        want = tr['post_%d' % i][fi]
        out = po(obs)
        if sp['kind'] == 'repaint':
          np.testing.assert_array_equal(helpers.to_np(out.board), want)
          for c, layer in out.layers.items():
            np.testing.assert_array_equal(helpers.to_np(layer).astype(bool), want == ord(c))
          assert sorted(out.layers) == sorted((set(chr(c) for c in tr['chars']) - set(sp['mapping'])) | set(sp['mapping'].values()))
        else:
          out = helpers.to_np(out)
          assert out.dtype == want.dtype and out.shape == want.shape, (out.shape, want.shape)
          np.testing.assert_array_equal(out, want)
      fi += 1


@pytest.mark.gpu
def test_device_to_array_unmapped_character_raises_and_croppers_chain():
  from pycolab_amd import cropping
  from pycolab_amd.engine import Engine
  t = helpers.load_template('warehouse_L0')
  eng = Engine.from_template(t, batch=3)
  crop = cropping.FixedCropper((1, 1), 4, 5, pad_char='#')
  crop.set_engine(eng)
  obs = eng.its_showtime()[0]
  with pytest.raises(RuntimeError):
    bad = rendering.ObservationToArray({' ': 0, '#': 1})    # '.' and others unmapped
    bad(obs)              # batch > 1: no wait for the device ...
    bad.check_errors()    # ... the error surfaces here (or at a later call)
  # post-processing a cropped observation uses the cropper's planes
  cropped = crop.crop(obs)
  feats = helpers.to_np(rendering.ObservationToFeatureArray('#P')(cropped))
  assert feats.shape == (3, 2, 4, 5) and feats.dtype == np.float32
  np.testing.assert_array_equal(feats[:, 0], (helpers.to_np(cropped.board) == ord('#')).astype(np.float32))
  # a repainted observation is a source too: repainter -> array / feature array (rendering.py:304-406 chained)
  repainted = rendering.ObservationCharacterRepainter({c: 'x' for c in '12345'})(obs)
  board = helpers.to_np(repainted.board)
  assert set(repainted.layers) == set(' #.xPX_') and (board == ord('x')).sum() == 3 * 5
  arr = helpers.to_np(rendering.ObservationToArray({' ': 0, '#': 1, '.': 2, 'x': 3, 'P': 4, 'X': 5, '_': 6}, dtype=np.int32)(repainted))
  lut = np.zeros(128, np.int32)
  for i, ch in enumerate(' #.xPX_'):
    lut[ord(ch)] = i
  np.testing.assert_array_equal(arr, lut[board])
  f2 = helpers.to_np(rendering.ObservationToFeatureArray('x_', permute=(1, 2, 0))(repainted))
  assert f2.shape == (3, 11, 10, 2)
  np.testing.assert_array_equal(f2[..., 0], (board == ord('x')).astype(np.float32))
  one = Engine.from_template(t, batch=1)
  obs1 = one.its_showtime()[0]
  with pytest.raises(RuntimeError):   # batch 1 raises at once, as the reference does
    rendering.ObservationToArray({' ': 0, '#': 1})(obs1)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['marauders', 'warehouse_L0', 'hello_world', 'scrolly_maze_L0'])
def test_device_feature_and_value_arrays_in_every_axis_order(name):
  """Every `permute` of ObservationToFeatureArray / ObservationToArray equals
  np.transpose of the default order (rendering.py:520-542, 640-661); the
  default and the channels-last orders take vector-store paths when the board
  is a whole number of dwords (marauders), element stores otherwise."""
  import itertools
  from pycolab_amd.engine import Engine
  t = helpers.load_template(name)
  eng = Engine.from_template(t, batch=70, auto_reset=True, seed=3)
  eng.its_showtime()
  eng.step_hashed(11, 0, 25)
  obs = eng.play(np.zeros(70, np.int32))[0]
  chars = ''.join(chr(c) for c in t.chars)
  base = helpers.to_np(rendering.ObservationToFeatureArray(chars)(obs))
  board = helpers.to_np(obs.board)
  for k, ch in enumerate(chars):
    np.testing.assert_array_equal(base[:, k], (board == ord(ch)).astype(np.float32))
  values = {ch: (i, 3 * i + 1, 250 - i) for i, ch in enumerate(chars)}
  rgb = helpers.to_np(rendering.ObservationToArray(values, dtype=np.uint8)(obs))
  lut = np.zeros((128, 3), np.uint8)
  for ch, v in values.items():
    lut[ord(ch)] = v
  np.testing.assert_array_equal(rgb, np.transpose(lut[board], (0, 3, 1, 2)))
  for perm in itertools.permutations(range(3)):
    got = helpers.to_np(rendering.ObservationToFeatureArray(chars, permute=perm)(obs))
    np.testing.assert_array_equal(got, np.transpose(base, (0,) + tuple(1 + a for a in perm)), err_msg=str(perm))
    got = helpers.to_np(rendering.ObservationToArray(values, dtype=np.uint8, permute=perm)(obs))
    np.testing.assert_array_equal(got, np.transpose(rgb, (0,) + tuple(1 + a for a in perm)), err_msg=str(perm))


@pytest.mark.gpu
@pytest.mark.parametrize('batch', [700, 82000])
def test_feature_array_fused_into_the_scrolly_maze_kernel(batch):
  """The headline game's kernel carries the same epilogue, in its cooperative
  (small batch) and single-wave (large batch) launch shapes."""
  import torch
  from pycolab_amd.engine import Engine
  t = helpers.load_template('scrolly_maze_L0')
  eng = Engine.from_template(t, batch=batch, auto_reset=True)
  eng.its_showtime()
  chars = '@P#a '
  fused = rendering.ObservationToFeatureArray(chars)
  assert fused.fuse_into(eng)
  for step in range(12):
    obs = eng.play(torch.randint(0, 5, (batch,), dtype=torch.int32, device='cuda'))[0]
    got = fused(obs)
    for k, ch in enumerate(chars):
      assert torch.equal(got[:, k], (obs.board == ord(ch)).to(torch.float32)), (step, ch)
    for ch, layer in obs.layers.items():
      assert torch.equal(layer, (obs.board == ord(ch)).to(torch.uint8))
  eng.step_n(torch.randint(0, 5, (5, batch), dtype=torch.int32, device='cuda'))  # no multi-step instance: falls back to single steps
  obs = eng.play(None)[0]
  got = fused(obs)
  assert torch.equal(got[:, 1], (obs.board == ord('P')).to(torch.float32))
  assert fused.fuse_into(eng, skip_layers=True)
  obs = eng.play(torch.randint(0, 5, (batch,), dtype=torch.int32, device='cuda'))[0]
  got = fused(obs)
  for k, ch in enumerate(chars):
    assert torch.equal(got[:, k], (obs.board == ord(ch)).to(torch.float32)), ch


@pytest.mark.gpu
@pytest.mark.parametrize('skip_layers', [False, True])
def test_feature_array_fused_into_the_step_kernel(skip_layers):
  """ObservationToFeatureArray.fuse_into(engine): the step kernel's render loop
  writes the float32 stack itself.  Same values as the separate
  post-processor kernel, every step, through auto-resets and with finished
  environments left frozen; with skip_layers the board stays right too."""
  import torch
  from pycolab_amd.engine import Engine
  t = helpers.load_template('marauders')
  B = 300
  eng = Engine.from_template(t, batch=B, auto_reset=True, seed=5)
  eng.its_showtime()
  eng.step_hashed(3, 0, 40)
  chars = 'PXB aqz'  # 'q' is no character of the game: a plane of zeros
  fused = rendering.ObservationToFeatureArray(chars)
  assert fused.fuse_into(eng, skip_layers=skip_layers)
  plain = rendering.ObservationToFeatureArray(chars)
  rng = np.random.RandomState(1)
  for step in range(60):
    eng._auto_reset = step % 4 != 3   # every fourth step leaves finished environments untouched
    obs = eng.play(rng.randint(0, 4, size=B).astype(np.int32))[0]
    got = fused(obs)
    assert isinstance(got, torch.Tensor) and got.shape == (B, len(chars), t.rows, t.cols)
    board = helpers.to_np(obs.board)
    for k, ch in enumerate(chars):
      np.testing.assert_array_equal(helpers.to_np(got[:, k]), (board == ord(ch)).astype(np.float32), err_msg='step %d %r' % (step, ch))
    if not skip_layers:
      np.testing.assert_array_equal(helpers.to_np(got), helpers.to_np(plain(obs)))
  assert helpers.to_np(eng.game_over).any() or True
  # an epilogue kind the engine's kernel does not write is refused and changes nothing (the table-driven kernel:
  # planar feature arrays only -- test_table_driven_kernel_writes_the_feature_array)
  tb = Engine.from_template(helpers.load_template('walkers_room'), batch=8)
  tb.its_showtime()
  assert not rendering.ObservationToFeatureArray('w', permute=(1, 2, 0)).fuse_into(tb)
  assert not rendering.ObservationToArray({c: 1.0 for c in (chr(x) for x in tb.template.chars)}, dtype=np.float32).fuse_into(tb)


@pytest.mark.gpu
@pytest.mark.parametrize('batch', [300, 40000])
@pytest.mark.parametrize('name,chars,n_actions', [('hello_world', '@1 #3', 4), ('hello_custom_A', '@2#', 4),
                                                   ('warehouse_L0', '#P_1X ', 5), ('warehouse_L2', 'P23X', 5),
                                                   ('better_scrolly_maze_L0', 'P@#a ', 5), ('better_scrolly_maze_L1', '@cP#', 5)])
def test_feature_array_fused_on_boards_of_any_size(name, chars, n_actions, batch):
  """The epilogue in pcx_hello_world_step, pcx_warehouse_step and pcx_better_scrolly_step, in both launch
  shapes, including boards that are not a whole number of dwords (110, 143, 4005 and 870 cells: the last
  dword of every float plane is written cell by cell)."""
  import torch
  from pycolab_amd.engine import Engine
  t = helpers.load_template(name)
  eng = Engine.from_template(t, batch=batch, auto_reset=True, seed=2)
  eng.its_showtime()
  eng.step_hashed(5, 0, 10)
  fused = rendering.ObservationToFeatureArray(chars)
  assert fused.fuse_into(eng), name
  guard = fused._fused[1]
  for step in range(6):
    obs = eng.play(torch.randint(0, n_actions, (batch,), dtype=torch.int32, device='cuda'))[0]
    got = fused(obs)
    assert got is guard and got.shape == (batch, len(chars), t.rows, t.cols)
    for k, ch in enumerate(chars):
      assert torch.equal(got[:, k], (obs.board == ord(ch)).to(torch.float32)), (name, step, ch)
    for ch, layer in obs.layers.items():
      assert torch.equal(layer, (obs.board == ord(ch)).to(torch.uint8)), (name, step, ch)
  eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize('skip_layers', [False, True])
@pytest.mark.parametrize('name,chars,n_actions,batch', [('scrolly_maze_L0', 'P@#a +', 5, 300), ('scrolly_maze_L0', 'abcP@# ', 5, 3000),
                                                         ('scrolly_maze_L0', '@P#', 5, 70000),
                                                         ('marauders', 'PXB aqz', 4, 300), ('marauders', 'BX|^P ', 4, 33000),
                                                         ('hello_world', '@1 #3', 4, 300), ('hello_world', '1234@', 4, 70000),
                                                         ('hello_custom_A', '@2#', 4, 500), ('warehouse_custom_B', 'P1X #', 5, 700),
                                                         ('warehouse_custom_B', 'X_.', 5, 90000)])
def test_channels_last_feature_array_fused_into_the_step_kernel(name, chars, n_actions, batch, skip_layers):
  _check_channels_last(name, chars, n_actions, batch, skip_layers)


@pytest.mark.gpu
def test_channels_last_feature_array_in_the_scrolly_maze_mask_path(monkeypatch):
  monkeypatch.setenv('PCX_SM_CODES', '0')  # the single-wave instance that composes from masks instead of owner codes
  _check_channels_last('scrolly_maze_L0', 'Pab@# ', 5, 70000, False)


def _check_channels_last(name, chars, n_actions, batch, skip_layers):
  """ObservationToFeatureArray(permute=(1, 2, 0)).fuse_into(engine): the render loop writes [B, rows, cols, depth]
  itself (the lanes of a wave exchange their layer dwords through LDS so that the stores stay contiguous): equal to
  the separate kernel every step, through auto-resets and with finished environments left frozen, in both launch
  shapes (scrolly_maze: cooperative, mask path and owner-code path); refused where the board is not a whole number
  of dwords."""
  import torch
  from pycolab_amd.engine import Engine
  t = helpers.load_template(name)
  assert (t.rows * t.cols) % 4 == 0
  eng = Engine.from_template(t, batch=batch, auto_reset=True, seed=5)
  eng.its_showtime()
  eng.step_hashed(3, 0, 12)
  fused = rendering.ObservationToFeatureArray(chars, permute=(1, 2, 0))
  assert fused.fuse_into(eng, skip_layers=skip_layers), name
  plain = rendering.ObservationToFeatureArray(chars, permute=(1, 2, 0))
  guard = fused._fused[1]
  for step in range(10 if batch > 5000 else 40):
    eng._auto_reset = step % 4 != 3   # every fourth step leaves finished environments untouched
    obs = eng.play(torch.randint(0, n_actions, (batch,), dtype=torch.int32, device='cuda'))[0]
    got = fused(obs)
    assert got is guard and got.shape == (batch, t.rows, t.cols, len(chars))
    for k, ch in enumerate(chars):
      assert torch.equal(got[..., k], (obs.board == ord(ch)).to(torch.float32)), (name, step, ch)
    if not skip_layers:
      assert torch.equal(got, plain(obs)), (name, step)
  eng.close()
  deep = Engine.from_template(helpers.load_template('marauders'), batch=64)   # 32 layers: the exchange areas would not fit LDS
  deep.its_showtime()
  many = ''.join(chr(c) for c in range(ord('A'), ord('A') + 32))
  assert not rendering.ObservationToFeatureArray(many, permute=(1, 2, 0)).fuse_into(deep)
  assert rendering.ObservationToFeatureArray(many).fuse_into(deep)
  deep.close()
  odd = Engine.from_template(helpers.load_template('better_scrolly_maze_L0'), batch=64)   # 4,005 cells
  odd.its_showtime()
  assert not rendering.ObservationToFeatureArray('P@', permute=(1, 2, 0)).fuse_into(odd)
  assert rendering.ObservationToFeatureArray('P@').fuse_into(odd)
  odd.close()


@pytest.mark.gpu
@pytest.mark.parametrize('skip_layers', [False, True])
@pytest.mark.parametrize('name,n_actions,batch,kind', [('scrolly_maze_L0', 5, 300, 'rgb'), ('scrolly_maze_L0', 5, 3000, 'scalar'),
                                                       ('scrolly_maze_L0', 5, 70000, 'rgb'), ('scrolly_maze_L0', 5, 70000, 'wide'),
                                                       ('marauders', 4, 300, 'scalar'), ('marauders', 4, 33000, 'rgb'),
                                                       ('hello_world', 4, 500, 'wide'), ('hello_world', 4, 70000, 'rgb'),
                                                       ('warehouse_custom_B', 5, 700, 'rgb'), ('warehouse_custom_B', 5, 90000, 'scalar')])
def test_value_array_fused_into_the_step_kernel(name, n_actions, batch, kind, skip_layers):
  """ObservationToArray.fuse_into(engine): the render loop sends every board dword through the value table (in LDS)
  and writes the array itself -- uint8 RGB vectors, float32 scalars, int64 vectors -- equal to the separate kernel
  every step, through auto-resets and with finished environments left frozen, in every launch shape; with
  skip_layers the step writes the board and the array only."""
  import torch
  from pycolab_amd.engine import Engine
  t = helpers.load_template(name)
  chars = [chr(c) for c in t.chars]
  rng = np.random.RandomState(4)
  if kind == 'rgb':
    mapping = {c: tuple(int(x) for x in rng.randint(0, 256, size=3)) for c in chars}
    args = dict(dtype=np.uint8)
  elif kind == 'scalar':
    mapping = {c: float(i) * 0.25 - 1.0 for i, c in enumerate(chars)}
    args = dict(dtype=np.float32)
  else:
    mapping = {c: tuple(int(x) for x in rng.randint(-2 ** 40, 2 ** 40, size=2)) for c in chars}
    args = dict(dtype=np.int64)
  mapping['~'] = mapping[chars[0]]  # (a key the game never shows)
  eng = Engine.from_template(t, batch=batch, auto_reset=True, seed=5)
  eng.its_showtime()
  eng.step_hashed(3, 0, 12)
  fused = rendering.ObservationToArray(mapping, **args)
  # (past the Infinity Cache the fused body loses to step + stand-alone kernel and fuse_into() declines by itself:
  # force=True is the A/B switch -- the body must be right at every size)
  assert fused.fuse_into(eng, skip_layers=skip_layers, force=True), name
  plain = rendering.ObservationToArray(mapping, **args)
  guard = fused._fused[1]
  for step in range(8 if batch > 5000 else 30):
    eng._auto_reset = step % 4 != 3   # every fourth step leaves finished environments untouched
    obs = eng.play(torch.randint(0, n_actions, (batch,), dtype=torch.int32, device='cuda'))[0]
    got = fused(obs)
    assert got is guard
    assert torch.equal(got, plain(obs)), (name, kind, step)
    if not skip_layers:
      for ch, layer in obs.layers.items():
        assert torch.equal(layer, (obs.board == ord(ch)).to(torch.uint8)), (name, step, ch)
  eng.check_errors()
  eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize('name,n_actions,batch', [('scrolly_maze_L0', 5, 300), ('scrolly_maze_L0', 5, 70000), ('marauders', 4, 2000),
                                                  ('hello_world', 4, 70000), ('warehouse_custom_B', 5, 900)])
def test_epilogue_without_any_plane(name, n_actions, batch):
  """skip_board=True: the step kernel writes the fused converter's array and nothing else of the observation
  (`board=None`).  Checked against a twin engine that runs the separate kernels on the same actions: a value
  array, then a feature stack, then channels last."""
  import torch
  from pycolab_amd.engine import Engine
  t = helpers.load_template(name)
  chars = [chr(c) for c in t.chars]
  mapping = {c: (3 * i + 1, 200 - i, i) for i, c in enumerate(chars)}
  makers = [lambda: rendering.ObservationToArray(mapping, dtype=np.uint8),
            lambda: rendering.ObservationToFeatureArray(''.join(chars[:5])),
            lambda: rendering.ObservationToFeatureArray(''.join(chars[:4]), permute=(1, 2, 0))]
  a = Engine.from_template(t, batch=batch, auto_reset=True, seed=6)
  b = Engine.from_template(t, batch=batch, auto_reset=True, seed=6)
  a.its_showtime(); b.its_showtime()
  for make in makers:
    fused, plain = make(), make()
    assert fused.fuse_into(a, skip_board=True), name
    for step in range(6):
      acts = torch.randint(0, n_actions, (batch,), dtype=torch.int32, device='cuda')
      oa, ob = a.play(acts)[0], b.play(acts)[0]
      assert oa.board is None
      assert torch.equal(fused(oa), plain(ob)), (name, step)
    fused.unfuse()
    oa, ob = a.play(acts)[0], b.play(acts)[0]   # the planes are written again
    assert torch.equal(oa.board, ob.board)
  a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['planes kept', 'skip_layers', 'skip_board'])
@pytest.mark.parametrize('name,n_actions,batch', [('scrolly_maze_L0', 5, 300), ('scrolly_maze_L0', 5, 70000), ('marauders', 4, 2000),
                                                  ('hello_world', 4, 70000), ('warehouse_custom_B', 5, 900)])
def test_repainter_fused_into_the_step_kernel(name, n_actions, batch, mode):
  """ObservationCharacterRepainter.fuse_into(engine): the step kernel writes the repainted board and its layers
  itself; still a planes source (a feature array chained behind it).  Against a twin engine that runs the separate
  kernels on the same actions."""
  import torch
  from pycolab_amd.engine import Engine
  t = helpers.load_template(name)
  chars = [chr(c) for c in t.chars]
  mapping = {chars[0]: '-', chars[1]: chars[2], chars[-1]: 'Z'}   # two characters merge, two get new ones
  a = Engine.from_template(t, batch=batch, auto_reset=True, seed=6)
  b = Engine.from_template(t, batch=batch, auto_reset=True, seed=6)
  a.its_showtime(); b.its_showtime()
  fused, plain = rendering.ObservationCharacterRepainter(mapping), rendering.ObservationCharacterRepainter(mapping)
  assert fused.fuse_into(a, skip_layers=mode == 'skip_layers', skip_board=mode == 'skip_board'), name
  outs = sorted((set(chars) - set(mapping)) | set(mapping.values()))
  feats_a, feats_b = rendering.ObservationToFeatureArray(''.join(outs)), rendering.ObservationToFeatureArray(''.join(outs))
  for step in range(6):
    acts = torch.randint(0, n_actions, (batch,), dtype=torch.int32, device='cuda')
    oa, ob = a.play(acts)[0], b.play(acts)[0]
    ra, rb = fused(oa), plain(ob)
    assert torch.equal(ra.board, rb.board), (name, step)
    assert sorted(ra.layers) == sorted(rb.layers) == outs
    for ch in outs:
      assert torch.equal(ra.layers[ch], rb.layers[ch]), (name, step, ch)
    assert torch.equal(feats_a(ra), feats_b(rb)), (name, step)   # the repainted planes chain as before
  a.close(); b.close()


@pytest.mark.gpu
def test_value_array_epilogue_refusals():
  from pycolab_amd.engine import Engine
  t = helpers.load_template('scrolly_maze_L0')
  eng = Engine.from_template(t, batch=64)
  eng.its_showtime()
  chars = [chr(c) for c in t.chars]
  full = {c: i for i, c in enumerate(chars)}
  assert not rendering.ObservationToArray({c: i for i, c in enumerate(chars[:-1])}, dtype=np.float32).fuse_into(eng)  # a character without a value
  assert not rendering.ObservationToArray(full, dtype=np.float32, permute=(1, 0)).fuse_into(eng)                    # a permuted axis order
  a = rendering.ObservationToArray(full, dtype=np.float32)
  assert a.fuse_into(eng)
  f = rendering.ObservationToFeatureArray(''.join(chars))
  assert f.fuse_into(eng) and a._fused is None   # the kernel feeds one array: the feature stack replaced the value array
  eng.close()
  odd = Engine.from_template(helpers.load_template('warehouse_L0'), batch=64)   # 110 cells
  odd.its_showtime()
  assert not rendering.ObservationToArray({chr(c): 1 for c in odd._template.chars}, dtype=np.uint8).fuse_into(odd)
  gen = Engine.from_template(helpers.load_template('walkers_room'), batch=64)    # the table-driven kernel
  gen.its_showtime()
  assert not rendering.ObservationToArray({chr(c): 1 for c in gen._template.chars}, dtype=np.uint8).fuse_into(gen)
  odd.close(); gen.close()


@pytest.mark.gpu
def test_fused_epilogue_belongs_to_the_engine():
  """The step kernel writes a fused converter's tensor at every step, so the ENGINE keeps
  converter and tensor alive (a converter dropped by the caller must not leave the kernel
  writing freed memory); a second converter takes the first one's place and the first runs
  as its own kernel again; unfuse() and close() stop the writes."""
  import gc
  import torch
  from pycolab_amd.engine import Engine
  t = helpers.load_template('marauders')
  B = 200
  eng = Engine.from_template(t, batch=B, auto_reset=True, seed=5)
  eng.its_showtime()
  first = rendering.ObservationToFeatureArray('PX')
  assert first.fuse_into(eng)
  kept = eng._epilogue[1]
  del first
  gc.collect()
  assert eng._epilogue is not None and eng._epilogue[1] is kept      # still owned: the kernel may write it
  junk = [torch.full((B, 2, t.rows, t.cols), 7.0, device='cuda') for _ in range(4)]  # would reuse freed blocks
  obs = eng.play(np.zeros(B, np.int32))[0]
  assert all(bool((x == 7.0).all()) for x in junk)
  assert torch.equal(kept[:, 0], (obs.board == ord('P')).to(torch.float32))
  a, b = rendering.ObservationToFeatureArray('PX'), rendering.ObservationToFeatureArray('B a')
  assert a.fuse_into(eng) and b.fuse_into(eng)                        # b replaces a in the kernel
  assert a._fused is None and b._fused is not None
  obs = eng.play(np.ones(B, np.int32))[0]
  board = obs.board
  for conv, chars in ((a, 'PX'), (b, 'B a')):                         # a: its own kernel again; b: fused
    got = conv(obs)
    for k, ch in enumerate(chars):
      assert torch.equal(got[:, k], (board == ord(ch)).to(torch.float32)), ch
  b.unfuse()
  assert eng._epilogue is None and b._fused is None
  stale = b(eng.play(np.ones(B, np.int32))[0])                        # separate kernel now
  assert torch.equal(stale[:, 0], (eng._result()[0].board == ord('B')).to(torch.float32))
  c = rendering.ObservationToFeatureArray('P')
  assert c.fuse_into(eng)
  eng.close()
  assert c._fused is None


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['planes kept', 'skip_layers', 'skip_board'])
@pytest.mark.parametrize('name', POSTED)
def test_fused_postprocessors_match_reference(name, mode):
  """The parity chain closed on FUSED outputs: the traces' recorded post-processors (outputs of the reference's own
  rendering.ObservationToArray / ObservationToFeatureArray / ObservationCharacterRepainter, rendering.py:340-661),
  each installed as the step kernel's epilogue with fuse_into() -- planes kept, uint8 layers dropped, every plane
  dropped -- must reproduce the reference's arrays at every recorded frame.  A converter the kernel cannot carry
  (fuse_into() answers False) is compared through the stand-alone kernel instead, so that nothing is skipped."""
  from pycolab_amd.engine import Engine
  tr = helpers.load_trace(name)
  specs, every = specs_of(tr)
  t = helpers.load_template(tr['template'])
  T, E = tr['actions'].shape
  fused_any = 0
  for i, sp in enumerate(specs):
    eng = Engine.from_template(t, batch=E, auto_reset=True, seed=helpers.GOLDEN_RNG_SEED)
    obs = eng.its_showtime()[0]
    po = make_post(sp)
    fused = po.fuse_into(eng, skip_layers=mode == 'skip_layers', skip_board=mode == 'skip_board')
    fused_any += bool(fused)
    if not fused and mode != 'planes kept':
      eng.close()
      continue  # (the stand-alone kernel is the 'planes kept' leg's business)
    fi = 0
    for step in range(T + 1):
      if step:
        obs = eng.play(tr['actions'][step - 1])[0]
      if step == 0 or step % every == 0:
        want = tr['post_%d' % i][fi]
        out = po(obs)
        where = '%s spec %d (%s) %s frame %d' % (name, i, sp['kind'], 'fused' if fused else 'stand-alone', step)
        if sp['kind'] == 'repaint':
          np.testing.assert_array_equal(helpers.to_np(out.board), want, err_msg=where)
          for c, layer in out.layers.items():
            np.testing.assert_array_equal(helpers.to_np(layer).astype(bool), want == ord(c), err_msg=where + ' layer ' + c)
        else:
          got = helpers.to_np(out)
          assert got.dtype == want.dtype and got.shape == want.shape, (where, got.shape, want.shape)
          np.testing.assert_array_equal(got, want, err_msg=where)
        fi += 1
    eng.close()
  assert fused_any, 'no recorded post-processor of %s could be fused' % name


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['planes kept', 'skip_layers', 'skip_board'])
@pytest.mark.parametrize('name,n_actions,batch', [('scrolly_maze_L0', 5, 300), ('scrolly_maze_L0', 5, 70000), ('scrolly_maze_L0', 5, 140000),
                                                  ('marauders', 4, 600), ('hello_world', 4, 400), ('warehouse_custom_B', 5, 700)])
def test_fused_outputs_match_the_numpy_oracle(name, n_actions, batch, mode):
  """Every fused converter against oracle/postprocess.py (the numpy restatement pinned on the reference's recorded
  outputs) instead of against another HIP kernel: value arrays, planar and channels-last feature stacks and the
  repainter, on the board the same actions produce (a twin engine's, so that the check also holds when the fused engine
  writes no board at all).  Environments are sampled at large batches; auto-resets and frozen environments included."""
  import torch
  from pycolab_amd.engine import Engine
  t = helpers.load_template(name)
  chars = [chr(c) for c in t.chars]
  rng = np.random.RandomState(9)
  rgb = {c: tuple(int(x) for x in rng.randint(0, 256, size=3)) for c in chars}
  scal = {c: float(i) * 0.5 - 2.0 for i, c in enumerate(chars)}
  remap = {chars[0]: '-', chars[1]: chars[2], chars[-1]: 'Z'}
  sel = ''.join(chars[:5])
  kinds = [('rgb', lambda: rendering.ObservationToArray(rgb, dtype=np.uint8), lambda b: opost.to_array(b, rgb, np.uint8)),
           ('scalar', lambda: rendering.ObservationToArray(scal, dtype=np.float32), lambda b: opost.to_array(b, scal, np.float32)),
           ('features', lambda: rendering.ObservationToFeatureArray(sel),
            lambda b: opost.feature_array({c: b == ord(c) for c in chars}, list(sel), b.shape)),
           ('features hwc', lambda: rendering.ObservationToFeatureArray(sel, permute=(1, 2, 0)),
            lambda b: opost.feature_array({c: b == ord(c) for c in chars}, list(sel), b.shape, (1, 2, 0))),
           ('repaint', lambda: rendering.ObservationCharacterRepainter(remap), lambda b: opost.repaint(b, chars, remap))]
  sample = np.unique(np.concatenate([np.arange(min(batch, 96)), rng.randint(0, batch, size=96), np.arange(max(0, batch - 96), batch)]))
  a = Engine.from_template(t, batch=batch, auto_reset=True, seed=6)
  b = Engine.from_template(t, batch=batch, auto_reset=True, seed=6)
  a.its_showtime(); b.its_showtime()
  a.step_hashed(3, 0, 10); b.step_hashed(3, 0, 10)
  checked = 0
  for kind, make, oracle in kinds:
    po = make()
    if not po.fuse_into(a, skip_layers=mode == 'skip_layers', skip_board=mode == 'skip_board'):
      continue
    for step in range(5):
      a._auto_reset = b._auto_reset = step % 3 != 2
      acts = torch.randint(0, n_actions, (batch,), dtype=torch.int32, device='cuda')
      oa, ob = a.play(acts)[0], b.play(acts)[0]
      out = po(oa)
      boards = helpers.to_np(ob.board[torch.from_numpy(sample).to(ob.board.device)]) if batch > 2000 else helpers.to_np(ob.board)[sample]
      idx = torch.from_numpy(sample).to('cuda')
      if kind == 'repaint':
        got_board = helpers.to_np(out.board[idx])
        got_layers = {c: helpers.to_np(l[idx]) for c, l in out.layers.items()}
        for j in range(len(sample)):
          wb, wl = oracle(boards[j])
          np.testing.assert_array_equal(got_board[j], wb, err_msg='%s %s step %d env %d' % (name, kind, step, sample[j]))
          assert sorted(got_layers) == sorted(wl)
          for c in wl:
            np.testing.assert_array_equal(got_layers[c][j].astype(bool), wl[c], err_msg='%s %s layer %s' % (name, kind, c))
      else:
        got = helpers.to_np(out[idx])
        for j in range(len(sample)):
          want = oracle(boards[j])
          assert got[j].dtype == want.dtype and got[j].shape == want.shape, (kind, got[j].shape, want.shape)
          np.testing.assert_array_equal(got[j], want, err_msg='%s %s step %d env %d' % (name, kind, step, sample[j]))
      checked += 1
    po.unfuse()
  assert checked, 'nothing could be fused into %s' % name
  a.close(); b.close()


@pytest.mark.gpu
def test_device_outputs_cross_dlpack_without_a_copy():
  """observations, cropped observations and feature arrays are handed on "via DLPack": the device
  tensors the engine returns export through the DLPack protocol and come back as views of the SAME device memory
  (no copy, no synchronisation) -- the board, a cropper's planes, a stand-alone and a fused feature array."""
  import torch
  from torch.utils import dlpack
  from pycolab_amd import cropping
  from pycolab_amd.engine import Engine
  t = helpers.load_template('scrolly_maze_L0')
  eng = Engine.from_template(t, batch=512, auto_reset=True)
  crop = cropping.ScrollingCropper(rows=5, cols=11, to_track=['P'], pad_char=' ')
  crop.set_engine(eng)
  obs = eng.its_showtime()[0]
  obs = eng.play(torch.randint(0, 5, (512,), dtype=torch.int32, device='cuda'))[0]
  cropped = crop.crop(obs)
  feats = rendering.ObservationToFeatureArray('P@#')(obs)
  fused = rendering.ObservationToFeatureArray('P@# ')
  assert fused.fuse_into(eng)
  obs2 = eng.play(torch.randint(0, 5, (512,), dtype=torch.int32, device='cuda'))[0]
  for what, ten in (('board', obs2.board), ('layer', obs2.layers['P']), ('cropped board', cropped.board),
                    ('feature array', feats), ('fused feature array', fused(obs2))):
    assert isinstance(ten, torch.Tensor) and ten.is_cuda, what
    back = torch.from_dlpack(dlpack.to_dlpack(ten))
    assert back.data_ptr() == ten.data_ptr() and back.shape == ten.shape and back.dtype == ten.dtype, what
    assert back.stride() == ten.stride(), what
    back2 = torch.from_dlpack(ten)  # (the __dlpack__ protocol a consumer framework calls)
    assert back2.data_ptr() == ten.data_ptr(), what
    assert torch.equal(back, ten)
  eng.close()


# ---- the table-driven kernel's feature-array epilogue (round 4) --------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize('jit', ['0', '1'])
@pytest.mark.parametrize('mode', ['planes kept', 'skip_layers', 'skip_board'])
@pytest.mark.parametrize('name', ['warehouse_L1'] + POSTED_TABLE_DRIVEN)  # (the traces with a planar feature array among their recorded post-processors)
def test_table_driven_kernel_feature_array_matches_reference(name, mode, jit, monkeypatch, tmp_path):
  """pcx_generic_step (both builds: table-driven and specialised per template) writes rendering.ObservationToFeatureArray
  in its default axis order from its render loop: the traces' recorded feature arrays (outputs of the reference's own
  class, rendering.py:545-661) at every recorded frame, with the uint8 planes kept, the layers dropped, everything dropped."""
  from pycolab_amd import _native as N
  from pycolab_amd.engine import Engine
  monkeypatch.setenv('PCX_FORCE_GENERIC', '1')
  monkeypatch.setenv('PCX_GENERIC_JIT', jit)
  monkeypatch.setenv('PCX_JIT_CACHE', str(tmp_path))
  tr = helpers.load_trace(name)
  specs, every = specs_of(tr)
  t = helpers.load_template(tr['template'])
  T, E = tr['actions'].shape
  seen = 0
  for i, sp in enumerate(specs):
    if sp['kind'] != 'features' or tuple(sp['permute'] or (0, 1, 2)) != (0, 1, 2):
      continue
    eng = Engine.from_template(t, batch=E, auto_reset=True, seed=helpers.GOLDEN_RNG_SEED)
    obs = eng.its_showtime()[0]
    assert N.lib().pcx_engine_kernel_name(eng._native).decode() == 'pcx_generic_step'
    assert int(N.lib().pcx_engine_launch_shape(eng._native)) == (31 if jit == '1' else 30)
    po = make_post(sp)
    assert po.fuse_into(eng, skip_layers=mode == 'skip_layers', skip_board=mode == 'skip_board'), (name, i)
    fi = 0
    for step in range(T + 1):
      if step:
        obs = eng.play(tr['actions'][step - 1])[0]
      if step == 0 or step % every == 0:
        want = tr['post_%d' % i][fi]
        got = helpers.to_np(po(obs))
        assert got.dtype == want.dtype and got.shape == want.shape
        np.testing.assert_array_equal(got, want, err_msg='%s spec %d frame %d (%s)' % (name, i, step, mode))
        if mode != 'skip_board':
          np.testing.assert_array_equal(helpers.to_np(obs.board), tr['boards'][step], err_msg='board, frame %d' % step)
        fi += 1
    eng.close()
    seen += 1
  assert seen, 'no planar feature array among the recorded post-processors of %s' % name


@pytest.mark.gpu
@pytest.mark.parametrize('jit', ['0', '1'])
@pytest.mark.parametrize('name,chars,batch', [('walkers_scroll_groups', None, 777), ('directives_z_order', None, 500), ('marauders_custom_A', 'PXB ab~', 333),
                                               ('walkers_room', None, 4300), ('warehouse_custom_B', None, 650)])
def test_table_driven_kernel_writes_the_feature_array(name, chars, batch, jit, monkeypatch, tmp_path):
  """... on games only that kernel steps (prefab walkers in scrolling groups, z-order directives, an unshipped marauders
  board) and on boards that are no whole number of dwords (7x11, 3x7): equal to the numpy oracle's feature array of the
  board (oracle/postprocess.py) and to the stand-alone kernel every step, through auto-resets, with finished environments
  left frozen; the three plane modes; a character the game does not have gives a plane of zeros; croppers and the epilogue
  exclude each other."""
  import torch
  from pycolab_amd import _native as N
  from pycolab_amd import cropping
  from pycolab_amd.engine import Engine
  monkeypatch.setenv('PCX_FORCE_GENERIC', '1')
  monkeypatch.setenv('PCX_GENERIC_JIT', jit)
  monkeypatch.setenv('PCX_JIT_CACHE', str(tmp_path))
  t = helpers.load_template(name)
  game_chars = [chr(c) for c in t.chars]
  chars = list(chars) if chars else game_chars[::-1] + ['~']
  n_act = max(1, int(t.n_actions))
  for mode in ('planes kept', 'skip_layers', 'skip_board'):
    eng = Engine.from_template(t, batch=batch, auto_reset=True, seed=9)
    twin = Engine.from_template(t, batch=batch, auto_reset=True, seed=9)
    eng.its_showtime(); twin.its_showtime()
    assert N.lib().pcx_engine_kernel_name(eng._native).decode() == 'pcx_generic_step'
    assert int(N.lib().pcx_engine_launch_shape(eng._native)) == (31 if jit == '1' else 30)
    fused, plain = rendering.ObservationToFeatureArray(chars), rendering.ObservationToFeatureArray(chars)
    assert fused.fuse_into(eng, skip_layers=mode == 'skip_layers', skip_board=mode == 'skip_board')
    rng = np.random.RandomState(2)
    for step in range(40):
      eng._auto_reset = twin._auto_reset = step % 4 != 3  # every fourth step leaves finished environments untouched
      a = rng.randint(0, n_act, size=batch).astype(np.int32)
      eng.step(a); twin.step(a)  # (step(): random actions make these games raise scrolling.Error now and then -- not this test's business)
      obs, ref = eng._result()[0], twin._result()[0]
      got = helpers.to_np(fused(obs))
      board = helpers.to_np(ref.board)
      layers = {c: helpers.to_np(ref.layers[c]) for c in game_chars}  # (occluded layers: layers[c] == (board == c))
      want = np.stack([opost.feature_array({c: layers[c][e] for c in game_chars}, chars, board.shape[1:]) for e in range(batch)])
      np.testing.assert_array_equal(got, want, err_msg='%s %s step %d' % (name, mode, step))
      np.testing.assert_array_equal(got, helpers.to_np(plain(ref)), err_msg='stand-alone kernel, step %d' % step)
      if mode != 'skip_board':
        np.testing.assert_array_equal(helpers.to_np(obs.board), board)
      if mode == 'planes kept':
        for c in game_chars:
          np.testing.assert_array_equal(helpers.to_np(obs.layers[c]), helpers.to_np(ref.layers[c]))
    assert got[:, chars.index('~')].sum() == 0
    if mode == 'planes kept':  # one or the other: croppers are refused while the epilogue is installed, taken once it is gone
      crop = cropping.ScrollingCropper(3, 3, [chr(t.z_order[-1])], scroll_margins=(1, 1))
      assert cropping.fuse_croppers(eng, [crop]) is False and not crop._fused
      fused.unfuse()
      assert cropping.fuse_croppers(eng, [crop]) is True
      assert not rendering.ObservationToFeatureArray(chars).fuse_into(eng)
    eng.close(); twin.close()
