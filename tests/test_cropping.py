"""Croppers: the oracle (CPU) and the device kernels (GPU) against cropped
observations recorded from the reference's own croppers (oracle/gen_golden.py),
plus the constructor guards of cropping.py."""
import json

import os

import numpy as np
import pytest

from oracle import binding
from pycolab_amd import cropping
from tests import helpers

CROPPED = ['scrolly_maze_L0', 'warehouse_L1', 'marauders', 'better_scrolly_maze_L0', 'better_scrolly_maze_L1',
           'better_scrolly_maze_L2', 'warehouse_custom_C', 'better_scrolly_custom_A', 'better_scrolly_custom_B', 'better_scrolly_custom_C', 'better_scrolly_custom_D']


def specs_of(trace):
  return json.loads(bytes(trace['crop_specs']).decode())


@pytest.mark.parametrize('name', CROPPED)
def test_oracle_croppers_match_reference(name):
  tr = helpers.load_trace(name)
  t = helpers.load_template(tr['template'])
  t.param[0] = helpers.GOLDEN_RNG_SEED
  T, E = tr['actions'].shape
  eng = binding.OracleEngine(t, E)
  crops = [binding.OracleCropper(eng, cropping.cropper_from_spec(sp)) for sp in specs_of(tr)]
  chars = list(tr['chars'])
  eng.reset()
  for step in range(T + 1):
    if step:
      eng.step(tr['actions'][step - 1], auto_reset=True)
    for i, cr in enumerate(crops):
      planes, err = cr.crop()
      assert not err.any()
      want = tr['crop_%d' % i][step]
      np.testing.assert_array_equal(planes, helpers.expected_planes(want, chars), err_msg='cropper %d frame %d' % (i, step))


def test_constructor_guards():
  """cropping.py:343-358, 381-390."""
  with pytest.raises(ValueError):
    cropping.ScrollingCropper(4, 5, ['P'], scroll_margins=(None, 1))    # even rows, egocentric
  with pytest.raises(ValueError):
    cropping.ScrollingCropper(5, 5, ['P'], scroll_margins=(3, 1))       # margins reach the centre
  c = cropping.ScrollingCropper(5, 7, ['P'], scroll_margins=(None, None))
  assert c._scroll_margins == (2, 3) and c.rows == 5 and c.cols == 7

  class Small(object):
    rows, cols = 4, 4
  with pytest.raises(ValueError):
    cropping.ScrollingCropper(5, 5, ['P']).set_engine(Small())          # bigger than the board, no pad
  ident = cropping.ObservationCropper()
  assert ident.crop('anything') == 'anything'


@pytest.mark.gpu
@pytest.mark.parametrize('name', CROPPED)
def test_device_croppers_match_reference(name):
  from pycolab_amd.engine import Engine
  tr = helpers.load_trace(name)
  t = helpers.load_template(tr['template'])
  T, E = tr['actions'].shape
  eng = Engine.from_template(t, batch=E, auto_reset=True, seed=helpers.GOLDEN_RNG_SEED)
  crops = [cropping.cropper_from_spec(sp) for sp in specs_of(tr)]
  for cr in crops:
    cr.set_engine(eng)
  chars = list(tr['chars'])
  obs = eng.its_showtime()[0]
  for step in range(T + 1):
    if step:
      obs = eng.play(tr['actions'][step - 1])[0]
    for i, cr in enumerate(crops):
      out = cr.crop(obs)
      want = tr['crop_%d' % i][step]
      np.testing.assert_array_equal(helpers.to_np(out.board), want, err_msg='cropper %d frame %d board' % (i, step))
      for k, ch in enumerate(chars):
        np.testing.assert_array_equal(helpers.to_np(out.layers[chr(ch)]).astype(np.uint8), (want == ch).astype(np.uint8))


@pytest.mark.gpu
def test_device_cropper_without_pad_raises_outside():
  from pycolab_amd.engine import Engine
  t = helpers.load_template('warehouse_L0')
  eng = Engine.from_template(t, batch=4)
  cr = cropping.FixedCropper((8, 8), 5, 5)  # leaves the 11x10 board, no pad character
  cr.set_engine(eng)
  obs = eng.its_showtime()[0]
  with pytest.raises(RuntimeError):
    cr.crop(obs)          # batch > 1: crop() does not wait for the device ...
    cr.check_errors()     # ... the error surfaces here (or at a later crop())
  with pytest.raises(RuntimeError):
    for _ in range(3):    # the asynchronous poll of an earlier crop() reports it
      cr.crop(obs)
      eng._bufs['done'].numpy()  # let the device catch up
  one = Engine.from_template(t, batch=1)
  cr1 = cropping.FixedCropper((8, 8), 5, 5)
  cr1.set_engine(one)
  obs1 = one.its_showtime()[0]
  with pytest.raises(RuntimeError):   # batch 1 raises at once, as the reference does
    cr1.crop(obs1)
  with pytest.raises(ValueError):
    bad = cropping.FixedCropper((0, 0), 3, 3, pad_char='?')
    bad.set_engine(eng)
    bad.crop(obs)


@pytest.mark.gpu
def test_device_cropper_outputs_are_zero_copy_device_tensors_and_survive_a_new_engine():
  """batch > 1: crop() returns views of the cropper's own output tensor (no
  host copy); a post-processor cached on a cropper follows it to a new engine
  (`set_engine`), whose planes live elsewhere."""
  import torch
  from pycolab_amd import rendering
  from pycolab_amd.engine import Engine
  t = helpers.load_template('better_scrolly_maze_L0')
  crop = cropping.ScrollingCropper(7, 10, ['P'], pad_char='#', scroll_margins=(None, 3))  # 70 cells: pitch 72
  feats = rendering.ObservationToFeatureArray('#P@')
  for B in (5, 9):
    eng = Engine.from_template(t, batch=B, auto_reset=True)
    crop.set_engine(eng)
    obs = eng.its_showtime()[0]
    for step in range(6):
      out = crop.crop(obs)
      assert isinstance(out.board, torch.Tensor) and out.board.is_cuda and out.board.shape == (B, 7, 10)
      assert out.board.data_ptr() == crop._out.ptr   # plane 0 of environment 0 of the bound tensor
      f = feats(out)
      assert isinstance(f, torch.Tensor) and f.is_cuda and f.dtype == torch.float32 and f.shape == (B, 3, 7, 10)
      board = helpers.to_np(out.board)
      for i, ch in enumerate('#P@'):
        np.testing.assert_array_equal(helpers.to_np(f[:, i]), (board == ord(ch)).astype(np.float32))
        np.testing.assert_array_equal(helpers.to_np(out.layers[ch]), (board == ord(ch)).astype(np.uint8))
      # the window is the engine's board around the player
      full = helpers.to_np(obs.board)
      pr, pc = np.argwhere(full[0] == ord('P'))[0]
      assert (board[0] == ord('P')).sum() == 1
      wr, wc = np.argwhere(board[0] == ord('P'))[0]
      top, left = pr - wr, pc - wc
      ref = np.full((7, 10), ord('#'), np.uint8)
      for r in range(7):
        for c in range(10):
          if 0 <= top + r < full.shape[1] and 0 <= left + c < full.shape[2]:
            ref[r, c] = full[0, top + r, left + c]
      np.testing.assert_array_equal(board[0], ref)
      obs = eng.play(np.full((B,), step % 4, np.int32))[0]
    eng.close()


# ---- croppers fused into the step kernel (cropping.fuse_croppers, pcx_engine_fuse_croppers) ----

FUSABLE = ['scrolly_maze_L0', 'warehouse_L1', 'marauders', 'better_scrolly_maze_L0', 'better_scrolly_maze_L1', 'better_scrolly_maze_L2',
           'warehouse_custom_C', 'better_scrolly_custom_A', 'better_scrolly_custom_B', 'better_scrolly_custom_C', 'better_scrolly_custom_D']


def _fusable(specs, drapes):
  """Up to four croppers the step kernels can run themselves: fixed, or tracking sprites only."""
  picked = [i for i, sp in enumerate(specs) if sp['kind'] == 'fixed' or not (set(sp['to_track']) & drapes)]
  return picked[:4]


@pytest.mark.gpu
@pytest.mark.parametrize('only_crops', [False, True])
@pytest.mark.parametrize('name', FUSABLE)
def test_fused_croppers_match_reference(name, only_crops):
  """The reference's own cropped observations, from croppers the step kernel
  runs itself (requested before its_showtime(): frame 0 included); the croppers
  the kernel cannot run (drape trackers) stay stand-alone next to them."""
  from pycolab_amd.engine import Engine
  tr = helpers.load_trace(name)
  t = helpers.load_template(tr['template'])
  T, E = tr['actions'].shape
  eng = Engine.from_template(t, batch=E, auto_reset=True, seed=helpers.GOLDEN_RNG_SEED)
  specs = specs_of(tr)
  crops = [cropping.cropper_from_spec(sp) for sp in specs]
  drapes = {chr(d['ch']) for d in t.drapes}
  fused = _fusable(specs, drapes)
  if only_crops:  # the stand-alone croppers would read a stale observation
    crops = [crops[i] for i in fused]
    kept = fused
    fused = list(range(len(crops)))
  else:
    kept = list(range(len(crops)))
    for cr in crops:
      cr.set_engine(eng)
  assert cropping.fuse_croppers(eng, [crops[i] for i in fused], only_crops=only_crops) is None  # deferred to showtime
  chars = list(tr['chars'])
  obs = eng.its_showtime()[0]
  assert all(crops[i]._fused for i in fused), 'the step kernel of %s should run these croppers itself' % name
  assert eng._only_crops == only_crops
  for step in range(T + 1):
    if step:
      obs = eng.play(tr['actions'][step - 1])[0]
    if not only_crops:
      np.testing.assert_array_equal(helpers.to_np(obs.board), tr['boards'][step], err_msg='frame %d board' % step)
    for i, cr in enumerate(crops):
      out = cr.crop(obs)
      want = tr['crop_%d' % kept[i]][step]
      np.testing.assert_array_equal(helpers.to_np(out.board), want, err_msg='cropper %d frame %d board' % (kept[i], step))
      for k, ch in enumerate(chars):
        np.testing.assert_array_equal(helpers.to_np(out.layers[chr(ch)]).astype(np.uint8), (want == ch).astype(np.uint8),
                                      err_msg='cropper %d frame %d layer %r' % (kept[i], step, chr(ch)))


@pytest.mark.gpu
@pytest.mark.parametrize('shape', ['coop', 'single'])
@pytest.mark.parametrize('name,batch', [('better_scrolly_maze_L0', 3000), ('better_scrolly_maze_L2', 700), ('warehouse_L0', 5000),
                                        ('marauders', 1500), ('hello_world', 900), ('scrolly_maze_L0', 2500), ('scrolly_maze_L2', 800),
                                        ('scrolly_custom_B', 1100), ('scrolly_custom_D', 700)])
def test_fused_croppers_equal_stand_alone_croppers(name, batch, shape, monkeypatch):
  """Many groups (a ragged last one), both launch shapes, episodes ending and
  restarting: the windows the step kernel writes are the windows the
  stand-alone cropper kernels cut from the observation; fusing in mid-episode
  carries the windows over; releasing the croppers hands them back."""
  import torch
  from pycolab_amd.engine import Engine
  monkeypatch.setenv('PCX_COOP_BELOW', '1000000' if shape == 'coop' else '0')
  if name == 'marauders':
    monkeypatch.setenv('PCX_EM_WAVES', '4' if shape == 'coop' else '1')
  t = helpers.load_template(name)
  names = [chr(sp['ch']) for sp in t.sprites]
  track = ['P'] if 'P' in names else names[:1]
  R, C = t.rows, t.cols

  def make():
    return [cropping.ScrollingCropper(5, 7, track, pad_char=chr(t.chars[0]), scroll_margins=(1, 2)),
            cropping.ScrollingCropper(min(R, 7), min(C, 10), track, scroll_margins=(2, 3), initial_offset=(1, -2)),
            cropping.FixedCropper((R - 3, C - 5), 6, 9, pad_char=chr(t.chars[1])),
            cropping.ScrollingCropper(3, 3, track, pad_char=chr(t.chars[0]), scroll_margins=(None, None), saccade=False)]

  a = Engine.from_template(t, batch=batch, auto_reset=True, seed=7)
  b = Engine.from_template(t, batch=batch, auto_reset=True, seed=7)
  ca, cb = make(), make()
  for cr in ca:
    cr.set_engine(a)
  for cr in cb:
    cr.set_engine(b)
  cropping.fuse_croppers(a, ca)
  oa, ob = a.its_showtime()[0], b.its_showtime()[0]
  assert all(cr._fused for cr in ca) and not any(cr._fused for cr in cb)

  def same(step):
    assert torch.equal(oa.board, ob.board), 'step %d: boards differ' % step
    for i, (x, y) in enumerate(zip(ca, cb)):
      wx, wy = x.crop(oa), y.crop(ob)
      assert torch.equal(wx.board, wy.board), 'step %d cropper %d: boards differ' % (step, i)
      for ch in wy.layers:
        assert torch.equal(wx.layers[ch], wy.layers[ch]), 'step %d cropper %d layer %r' % (step, i, ch)

  same(0)
  n_act = int(t.n_actions)
  rng = np.random.RandomState(3)
  for step in range(1, 49):
    acts = rng.randint(0, n_act, size=batch).astype(np.int32)
    oa, ob = a.play(acts)[0], b.play(acts)[0]
    same(step)  # (every step: a stand-alone window only moves when crop() is called)
    if step == 20:  # b's croppers join its step kernel in mid-episode ...
      assert cropping.fuse_croppers(b, cb) is True
    if step == 30:  # ... and a's leave theirs
      assert cropping.fuse_croppers(a, []) is True
      assert not any(cr._fused for cr in ca)
  a.close()
  b.close()


@pytest.mark.gpu
@pytest.mark.parametrize('name,batch,kernel,shape', [
    (n, b, k, sh) for n, b, k in [('warehouse_L0', 1500, 'pcx_warehouse_step'), ('warehouse_custom_B', 700, 'pcx_warehouse_step'),
                                  ('marauders', 900, 'pcx_marauders_step'), ('hello_world', 600, 'pcx_hello_world_step'),
                                  ('better_scrolly_maze_L1', 500, 'pcx_better_scrolly_step'),
                                  ('better_scrolly_maze_L0', 300, 'pcx_better_scrolly_step'),   # 45 x 89: two 64-bit column halves
                                  ('better_scrolly_maze_L2', 300, 'pcx_better_scrolly_step'),   # 29 x 89
                                  ('better_scrolly_custom_B', 800, 'pcx_better_scrolly_step')] for sh in ('coop', 'single')] + [
    # pcx_scrolly_maze_step: its single-wave shapes (shipped 10x30 instance, run-time-shape instance)
    ('scrolly_maze_L0', 900, 'pcx_scrolly_maze_step', 'single'), ('scrolly_maze_L2', 400, 'pcx_scrolly_maze_step', 'single'),
    ('scrolly_custom_B', 700, 'pcx_scrolly_maze_step', 'single'), ('scrolly_custom_D', 500, 'pcx_scrolly_maze_step', 'coop'),
    # ... and its cooperative shape (round 4: windows that follow a drape move after the workgroup's export of the
    # curtains): 64, 32 and 16 environments per workgroup, the last with four lanes per environment in the logic phase
    ('scrolly_maze_L0', 900, 'pcx_scrolly_maze_step', 'coop'), ('scrolly_maze_L1', 20000, 'pcx_scrolly_maze_step', 'coop'),
    ('scrolly_maze_L2', 9000, 'pcx_scrolly_maze_step', 'coop'), ('scrolly_maze_L0', 8, 'pcx_scrolly_maze_step', 'coop')])
def test_hand_written_kernels_fuse_drape_tracking_croppers(name, batch, kernel, shape, monkeypatch):
  """A fused cropper may follow a drape in the hand-written kernels too (boards of at most 63 x 128 cells): the
  logic wave takes the median of the raw curtain it has just exported (pcx_stream.h curtain_centroid).  Windows
  equal the stand-alone cropper kernels' on a twin engine, every step, drapes emptying and refilling across
  episodes, priority lists that fall through to a sprite."""
  import torch
  from pycolab_amd import _native as N
  from pycolab_amd.engine import Engine
  monkeypatch.setenv('PCX_COOP_BELOW', '1000000' if shape == 'coop' else '0')
  if name == 'marauders':
    monkeypatch.setenv('PCX_EM_WAVES', '4' if shape == 'coop' else '1')
  t = helpers.load_template(name)
  sprites = [chr(sp['ch']) for sp in t.sprites]
  drapes = [chr(d['ch']) for d in t.drapes]
  pad = chr(t.chars[0])

  def make():
    return [cropping.ScrollingCropper(3, 5, drapes[:1], pad_char=pad, scroll_margins=(None, 1)),
            cropping.ScrollingCropper(min(t.rows, 5), min(t.cols, 8), drapes[-1:] + sprites[:1], scroll_margins=(1, 2),
                                      initial_offset=(0, 1), saccade=True),
            cropping.ScrollingCropper(4, 6, sprites[-1:] + drapes[:1], pad_char=pad, scroll_margins=(1, 1), saccade=False)]

  a = Engine.from_template(t, batch=batch, auto_reset=True, seed=11)
  b = Engine.from_template(t, batch=batch, auto_reset=True, seed=11)
  ca, cb = make(), make()
  for cr in ca:
    cr.set_engine(a)
  for cr in cb:
    cr.set_engine(b)
  assert cropping.fuse_croppers(a, ca) is None
  oa, ob = a.its_showtime()[0], b.its_showtime()[0]
  if os.environ.get('PCX_FORCE_GENERIC') != '1':  # (the whole suite is also run through the table-driven kernel)
    assert N.lib().pcx_engine_kernel_name(a._native).decode() == kernel
  assert all(cr._fused for cr in ca) and not any(cr._fused for cr in cb)
  n_act = max(1, int(t.n_actions))
  rng = np.random.RandomState(8)
  for step in range(0, 45):
    if step:
      acts = rng.randint(0, n_act, size=batch).astype(np.int32)
      a.step(acts); b.step(acts)
      oa, ob = a._result()[0], b._result()[0]
    assert torch.equal(oa.board, ob.board), 'step %d: boards differ' % step
    for i, (x, y) in enumerate(zip(ca, cb)):
      wx, wy = x.crop(oa), y.crop(ob)
      assert torch.equal(wx.board, wy.board), 'step %d cropper %d: windows differ' % (step, i)
      for ch in wy.layers:
        assert torch.equal(wx.layers[ch], wy.layers[ch]), 'step %d cropper %d layer %r' % (step, i, ch)
    if step == 18:  # b's croppers join its step kernel in mid-episode
      assert cropping.fuse_croppers(b, cb) is True
  a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize('build', helpers.BUILDS)
@pytest.mark.parametrize('waves', ['1', '4'])
@pytest.mark.parametrize('name,batch', [('walkers_room', 900), ('walkers_scroll_groups', 700), ('directives_z_order', 500),
                                        ('marauders', 600), ('warehouse_L0', 1000), ('hello_world', 400),
                                        ('better_scrolly_custom_B', 300)])
def test_table_driven_kernel_fuses_croppers_drape_trackers_included(name, batch, waves, build, monkeypatch):
  """pcx_generic_step runs croppers itself, also those that follow a drape
  (the median of its raw curtain, cropping.py:590-598) or a priority list of
  drapes and sprites: step by step the windows equal those of the stand-alone
  cropper kernels on a twin engine; joining and leaving mid-episode."""
  import torch
  from pycolab_amd.engine import Engine
  helpers.force_generic(monkeypatch, build)  # (both builds of the kernel: the one hiprtc makes for the template too)
  monkeypatch.setenv('PCX_GENERIC_WAVES', waves)
  t = helpers.load_template(name)
  sprites = [chr(sp['ch']) for sp in t.sprites]
  drapes = [chr(d['ch']) for d in t.drapes]
  R, C = t.rows, t.cols
  pad = chr(t.chars[0])

  def make():
    out = [cropping.ScrollingCropper(5, 7, sprites[:1], pad_char=pad, scroll_margins=(1, 2)),
           cropping.FixedCropper((R - 3, C - 5), 6, 9, pad_char=chr(t.chars[1]))]
    if drapes:
      out.append(cropping.ScrollingCropper(3, 5, drapes[:1], pad_char=pad, scroll_margins=(None, 1)))
      out.append(cropping.ScrollingCropper(min(R, 4), min(C, 6), drapes[-1:] + sprites[-1:], scroll_margins=(1, 2),
                                           initial_offset=(0, 1), saccade=True))
    else:
      out.append(cropping.ScrollingCropper(min(R, 4), min(C, 6), sprites[-1:] + sprites[:1], scroll_margins=(1, 2), saccade=False))
    return out

  a = Engine.from_template(t, batch=batch, auto_reset=True, seed=7)
  b = Engine.from_template(t, batch=batch, auto_reset=True, seed=7)
  ca, cb = make(), make()
  for cr in ca:
    cr.set_engine(a)
  for cr in cb:
    cr.set_engine(b)
  cropping.fuse_croppers(a, ca)
  oa, ob = a.its_showtime()[0], b.its_showtime()[0]
  from pycolab_amd import _native as N
  assert N.lib().pcx_engine_kernel_name(a._native).decode() == 'pcx_generic_step'
  assert all(cr._fused for cr in ca) and not any(cr._fused for cr in cb)

  def same(step):
    assert torch.equal(oa.board, ob.board), 'step %d: boards differ' % step
    for i, (x, y) in enumerate(zip(ca, cb)):
      wx, wy = x.crop(oa), y.crop(ob)
      assert torch.equal(wx.board, wy.board), 'step %d cropper %d: boards differ' % (step, i)
      for ch in wy.layers:
        assert torch.equal(wx.layers[ch], wy.layers[ch]), 'step %d cropper %d layer %r' % (step, i, ch)

  same(0)
  n_act = max(1, int(t.n_actions))
  rng = np.random.RandomState(5)
  for step in range(1, 41):
    acts = rng.randint(0, n_act, size=batch).astype(np.int32)
    a.step(acts); b.step(acts)  # (step(): random actions make walkers_scroll_groups raise scrolling.Error in some environments, as the reference would)
    oa, ob = a._result()[0], b._result()[0]
    same(step)
    if step == 15:
      assert cropping.fuse_croppers(b, cb) is True
    if step == 28:
      assert cropping.fuse_croppers(a, []) is True
  # windows only: the full-board planes are no longer written, the windows still are
  c = Engine.from_template(t, batch=batch, auto_reset=True, seed=7)
  cc = make()
  for cr in cc:
    cr.set_engine(c)
  cropping.fuse_croppers(c, cc, only_crops=True)
  oc = c.its_showtime()[0]
  d = Engine.from_template(t, batch=batch, auto_reset=True, seed=7)
  cd = make()
  for cr in cd:
    cr.set_engine(d)
  od = d.its_showtime()[0]
  for cr in cd:
    cr.crop(od)  # (the fused croppers cropped frame 0 in the showtime launch)
  rng = np.random.RandomState(5)
  for step in range(1, 13):
    acts = rng.randint(0, n_act, size=batch).astype(np.int32)
    c.step(acts); d.step(acts)
    oc, od = c._result()[0], d._result()[0]
    for i, (x, y) in enumerate(zip(cc, cd)):
      assert torch.equal(x.crop(oc).board, y.crop(od).board), 'windows only, step %d cropper %d' % (step, i)
  for e in (a, b, c, d):
    e.close()


@pytest.mark.gpu
def test_croppers_released_from_a_windows_only_fusion_keep_their_output_until_the_next_step():
  import torch
  from pycolab_amd.engine import Engine
  t = helpers.load_template('better_scrolly_maze_L1')
  eng = Engine.from_template(t, batch=200, auto_reset=True)
  cr = cropping.ScrollingCropper(7, 9, ['P'], pad_char='#', scroll_margins=(2, 3))
  cropping.fuse_croppers(eng, [cr], only_crops=True)
  obs = eng.its_showtime()[0]
  for step in range(5):
    obs = eng.play(np.full((200,), step % 4, np.int32))[0]
  want = cr.crop(obs).board.clone()
  assert cropping.fuse_croppers(eng, []) is True        # released: the full-board planes are stale until the next step
  assert torch.equal(cr.crop(obs).board, want)           # ... so the window the fusion wrote stands
  obs = eng.play(np.zeros((200,), np.int32))[0]          # planes written again: the stand-alone kernels take over
  ref = cropping.ScrollingCropper(7, 9, ['P'], pad_char='#', scroll_margins=(2, 3))
  got = cr.crop(obs).board
  full = helpers.to_np(obs.board)
  win = helpers.to_np(got)
  for b in (0, 57, 199):                                 # the window shows the player where the board has it
    (pr, pc), = np.argwhere(full[b] == ord('P'))
    (wr, wc), = np.argwhere(win[b] == ord('P'))
    top, left = pr - wr, pc - wc
    for r in range(7):
      for c in range(9):
        inside = 0 <= top + r < full.shape[1] and 0 <= left + c < full.shape[2]
        assert win[b, r, c] == (full[b, top + r, left + c] if inside else ord('#'))
  del ref
  eng.close()


@pytest.mark.gpu
def test_fuse_croppers_answers_false_where_the_kernel_cannot():
  from pycolab_amd.engine import Engine
  t = helpers.load_template('marauders_unoccluded')   # (table-driven kernel) unoccluded layers: the windows derive layers from the board
  eng = Engine.from_template(t, batch=8, auto_reset=True)
  cr = cropping.ScrollingCropper(5, 7, [chr(t.sprites[0]['ch'])], pad_char=chr(t.chars[0]), scroll_margins=(1, 2))
  cr.set_engine(eng)
  obs = eng.its_showtime()[0]
  assert cropping.fuse_croppers(eng, [cr]) is False and not cr._fused
  assert helpers.to_np(cr.crop(obs).board).shape == (8, 5, 7)   # still crops, as its own kernels
  um = Engine.from_template(helpers.load_template('scrolly_maze_L1_unoccluded'), batch=8, auto_reset=True)
  cu = cropping.ScrollingCropper(5, 11, ['P'], pad_char=' ', scroll_margins=(1, 2))
  cu.set_engine(um)
  um.its_showtime()
  assert cropping.fuse_croppers(um, [cu]) is False                  # unoccluded layers: the windows derive layers from the board
  # (round 4: drape trackers fuse on better_scrolly_maze's 89-column boards and in pcx_scrolly_maze_step's cooperative
  # shape too: test_hand_written_kernels_fuse_drape_tracking_croppers)
  t2 = helpers.load_template('warehouse_L0')
  eng2 = Engine.from_template(t2, batch=8, auto_reset=True)
  drape = cropping.ScrollingCropper(3, 3, ['X'], pad_char=' ', scroll_margins=(None, None))
  drape.set_engine(eng2)
  eng2.its_showtime()
  sprite = cropping.ScrollingCropper(3, 3, ['P'], pad_char=' ', scroll_margins=(None, None))
  with pytest.raises(ValueError):                                  # the stand-alone cropper would read stale planes
    cropping.fuse_croppers(eng2, [sprite], only_crops=True)


def _centroid_like_the_kernels(curtain):
  """pcx_stream.h curtain_centroid / pcx_generic.hip drape_centroid, restated step by step on a bool curtain: rows
  as one or two 64-bit column vectors (round 4: boards up to 128 columns), per-column counts kept bit-sliced in six
  planes per half (a ripple-carry add per row), the two middle order statistics by prefix counts, their mean
  truncated."""
  R, C = curtain.shape
  M64 = (1 << 64) - 1
  rows = [sum(1 << c for c in range(C) if curtain[r, c]) for r in range(R)]
  halves = 2 if C > 64 else 1
  n, half_planes = 0, [[0] * 6 for _ in range(halves)]
  for v in rows:
    for h in range(halves):
      carry = (v >> (64 * h)) & M64
      n += bin(carry).count('1')
      for k in range(6):
        half_planes[h][k], carry = half_planes[h][k] ^ carry, half_planes[h][k] & carry
  planes = [sum(half_planes[h][k] << (64 * h) for h in range(halves)) for k in range(6)]
  if n == 0:
    return None
  lo_rank, hi_rank = (n - 1) // 2, n // 2

  def middle(counts):
    seen, lo, hi = 0, -1, -1
    for i, cnt in enumerate(counts):
      if lo < 0 and seen + cnt > lo_rank:
        lo = i
      if hi < 0 and seen + cnt > hi_rank:
        hi = i
      seen += cnt
    return (lo + hi) >> 1
  col_counts = [sum(((planes[k] >> c) & 1) << k for k in range(6)) for c in range(C)]
  return middle([bin(v).count('1') for v in rows]), middle(col_counts)


def test_the_kernels_drape_median_is_the_references():
  """cropping.py:590-598: `tuple(int(np.median(dim)) for dim in curtain.nonzero())` -- the arithmetic the fused
  croppers use for it (no sort, no division) on random curtains up to the 63 x 128 limit, dense, sparse and empty."""
  rng = np.random.RandomState(12)
  for trial in range(500):
    R, C = int(rng.randint(1, 64)), int(rng.randint(1, 129))
    density = rng.choice([0.0, 0.02, 0.3, 0.9, 1.0])
    curtain = rng.rand(R, C) < density
    if trial % 7 == 0 and curtain.size > 1:
      curtain[:] = False
      curtain[rng.randint(R), rng.randint(C)] = True   # a single cell
    got = _centroid_like_the_kernels(curtain)
    want = tuple(int(np.median(dim)) for dim in curtain.nonzero()) if curtain.any() else None
    assert got == want, (R, C, density, got, want)


# ---- crop, THEN post-process (human_ui.py:252-265; better_scrolly_maze.py:237-247 into rendering.py:545-661) ----------
CROP_POSTED = ['better_scrolly_maze_L0', 'warehouse_L1', 'marauders']


def crop_post_specs_of(trace):
  return [(int(ci), sp) for ci, sp in json.loads(bytes(trace['crop_post_specs']).decode())]


def _post_frames(T, every):
  return [0] + [t for t in range(1, T + 1) if t % every == 0]


@pytest.mark.parametrize('name', CROP_POSTED)
def test_oracle_feature_stack_of_a_cropped_observation_matches_reference(name):
  """The numpy restatement of ObservationToFeatureArray (oracle/postprocess.py) on the reference's recorded CROPPED
  boards against what the reference's own post-processor made of its own cropper's output (the traces' crop_post_*)."""
  from oracle import postprocess as opost
  tr = helpers.load_trace(name)
  chars = [chr(c) for c in tr['chars']]
  T, E = tr['actions'].shape
  every = int(tr['post_every'][0])
  for i, (ci, sp) in enumerate(crop_post_specs_of(tr)):
    for fi, t in enumerate(_post_frames(T, every)):
      for e in range(0, E, 3):
        board = tr['crop_%d' % ci][t, e]
        got = opost.feature_array({c: board == ord(c) for c in chars}, list(sp['layers']), board.shape, sp['permute'])
        want = tr['crop_post_%d' % i][fi, e]
        assert got.shape == want.shape and got.dtype == want.dtype
        np.testing.assert_array_equal(got, want, err_msg='%s spec %d frame %d env %d' % (name, i, t, e))


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['planes kept', 'skip_layers', 'skip_board', 'only_crops + skip_board'])
@pytest.mark.parametrize('name', CROP_POSTED)
def test_fused_window_feature_stack_matches_reference(name, mode):
  """ObservationToFeatureArray.fuse_into(engine, source=cropper): the step kernel cuts the window AND writes its float32
  stack in the same launch (pcx_cropper_set_features).  Against the reference's cropper -> post-processor outputs
  recorded in the traces, frame 0 included, windows of 300, 70, 49, 27 and 99 cells (whole dwords and not), planar and
  channels last, with the window's uint8 planes kept / dropped, and with the full-board planes dropped as well."""
  from pycolab_amd import rendering
  from pycolab_amd.engine import Engine
  tr = helpers.load_trace(name)
  t = helpers.load_template(tr['template'])
  T, E = tr['actions'].shape
  every = int(tr['post_every'][0])
  specs = specs_of(tr)
  cps = crop_post_specs_of(tr)
  drapes = {chr(d['ch']) for d in t.drapes}
  for i, (ci, sp) in enumerate(cps):
    if any(c in drapes for c in specs[ci]['to_track']):
      continue  # (this kernel does not fuse a drape tracker: nothing to attach the stack to)
    eng = Engine.from_template(t, batch=E, auto_reset=True, seed=helpers.GOLDEN_RNG_SEED)
    cr = cropping.cropper_from_spec(specs[ci])
    only = mode.startswith('only_crops')
    assert cropping.fuse_croppers(eng, [cr], only_crops=only) is None  # deferred to showtime
    obs = eng.its_showtime()[0]
    assert cr._fused
    conv = rendering.ObservationToFeatureArray(list(sp['layers']), permute=sp['permute'])
    assert conv.fuse_into(eng, source=cr, skip_layers=mode == 'skip_layers', skip_board=mode.endswith('skip_board')), (name, i)
    fi = 0
    for step in range(T + 1):
      if step:
        obs = eng.play(tr['actions'][step - 1])[0]
      cropped = cr.crop(obs)
      if step == 0 or step % every == 0:
        got = helpers.to_np(conv(cropped))
        want = tr['crop_post_%d' % i][fi]
        assert got.shape == want.shape and got.dtype == want.dtype, (got.shape, want.shape)
        np.testing.assert_array_equal(got, want, err_msg='%s spec %d (%s) frame %d' % (name, i, mode, step))
        if mode == 'planes kept':  # the window's own planes are still the reference's cropped observation
          np.testing.assert_array_equal(helpers.to_np(cropped.board), tr['crop_%d' % ci][step])
        fi += 1
    conv.unfuse()
    obs = eng.play(tr['actions'][0])[0]
    np.testing.assert_array_equal(helpers.to_np(conv(cr.crop(obs))).shape, tr['crop_post_%d' % i][0].shape)  # stand-alone again
    eng.close()


@pytest.mark.gpu
def test_fused_window_feature_stack_refusals_and_big_batch():
  from pycolab_amd import rendering
  from pycolab_amd.engine import Engine
  import torch
  t = helpers.load_template('better_scrolly_maze_L0')
  eng = Engine.from_template(t, batch=70000, auto_reset=True, seed=3)
  cr = cropping.ScrollingCropper(rows=10, cols=30, to_track=['P'], scroll_margins=(2, 3), initial_offset=(-3, -9))
  loose = cropping.ScrollingCropper(rows=5, cols=5, to_track=['P'], pad_char=' ', scroll_margins=(1, 1))
  loose.set_engine(eng)
  eng.its_showtime()
  conv = rendering.ObservationToFeatureArray('P@#abc +')
  assert not conv.fuse_into(eng, source=cr)             # not fused (not even attached)
  assert cropping.fuse_croppers(eng, [cr])
  assert not conv.fuse_into(eng, source=loose)          # attached, but runs as its own kernel
  assert not rendering.ObservationToFeatureArray('PP').fuse_into(eng, source=cr)   # a layer twice
  assert conv.fuse_into(eng, source=cr)
  plain = rendering.ObservationToFeatureArray('P@#abc +')
  for step in range(6):  # the single-wave launch shape, 70,000 environments: fused == stand-alone on the same window
    obs = eng.play(torch.randint(0, 5, (70000,), dtype=torch.int32, device='cuda'))[0]
    cropped = cr.crop(obs)
    assert torch.equal(conv(cropped), plain(cropped)), step
  gen = Engine.from_template(helpers.load_template('walkers_room'), batch=64)   # the table-driven kernel
  g = cropping.FixedCropper((0, 0), 4, 4)
  gen.its_showtime()
  if cropping.fuse_croppers(gen, [g]):
    assert not rendering.ObservationToFeatureArray('w').fuse_into(gen, source=g)
  eng.close(); gen.close()
