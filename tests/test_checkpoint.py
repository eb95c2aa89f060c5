"""pcx_engine_export_state / _import_state (checkpoint/resume): an episode
resumed from a checkpoint in ANOTHER engine continues exactly as the engine that
made it -- every kernel family, RNG draw counters and sticky scroll state included."""
import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

CASES = [('scrolly_maze_L0', 700, 0), ('marauders', 500, 0), ('warehouse_L1', 600, 0), ('better_scrolly_maze_L1', 300, 0),
         ('hello_world', 400, 0), ('walkers_scroll_groups', 300, 0), ('directives_z_order', 300, 0), ('marauders', 300, 1),
         ('marauders', 300, 2), ('walkers_scroll_groups', 300, 2)]  # (2: through the table-driven kernel's build specialised for the template)


@pytest.mark.parametrize('name,batch,force_generic', CASES)
def test_resume_equals_uninterrupted(name, batch, force_generic, monkeypatch):
  from tests.hip_adapter import HipAdapter
  if force_generic:
    helpers.force_generic(monkeypatch, 'specialised' if force_generic == 2 else 'table-driven')
  t = helpers.load_template(name)
  a = HipAdapter(t, batch, seed=7)
  a.reset()
  a.step_hashed(0xC0FFEE, 0, 37)
  blob = a.eng.export_state(with_observation=True)
  assert blob.dtype == np.uint8 and blob.size > 64 * batch // 64
  b = HipAdapter(t, batch, seed=7)
  b.reset()
  b.step_hashed(0xBAD, 0, 5)              # somewhere else entirely
  b.eng.import_state(blob)
  for name_ in ('planes', 'reward', 'reward_set', 'discount', 'done', 'frame'):
    np.testing.assert_array_equal(b.read(name_), a.read(name_), err_msg='right after import: ' + name_)
  for chunk in range(3):
    auto = chunk != 1                      # one chunk leaves finished environments frozen (their frame must survive too)
    a.step_hashed(0xC0FFEE, 37 + 20 * chunk, 20, auto_reset=auto)
    b.step_hashed(0xC0FFEE, 37 + 20 * chunk, 20, auto_reset=auto)
    for name_ in ('planes', 'reward', 'reward_set', 'discount', 'done', 'frame', 'error'):
      np.testing.assert_array_equal(b.read(name_), a.read(name_), err_msg='chunk %d: %s' % (chunk, name_))
    np.testing.assert_array_equal(b.sprites(), a.sprites())
  other = HipAdapter(t, batch + 64, seed=7)
  other.reset()
  with pytest.raises(ValueError):          # a checkpoint of another batch is refused
    other.eng.import_state(blob)


@pytest.mark.parametrize('fused,only_crops,with_observation', [(False, False, True), (False, False, False), (True, False, True),
                                                               (True, False, False), (True, True, True)])
def test_resume_restores_cropper_windows(fused, only_crops, with_observation):
  """A ScrollingCropper's window is state too (cropping.py:393-426: it pans only when the tracked entity leaves the
  margins, so where it stands depends on the path): a checkpoint carries every attached cropper's corners -- and, with
  the observation, its cropped planes -- and the resumed engine's croppers continue exactly as the exporting one's,
  stand-alone or fused into the step kernel, instead of re-centring."""
  import torch
  from pycolab_amd import cropping
  from pycolab_amd.engine import Engine
  t = helpers.load_template('better_scrolly_maze_L1')
  B = 300

  def build(seed_steps):
    eng = Engine.from_template(t, batch=B, auto_reset=True, seed=7)
    crops = [cropping.ScrollingCropper(rows=7, cols=11, to_track=['P'], pad_char=' ', scroll_margins=(2, 3)),
             cropping.ScrollingCropper(rows=5, cols=9, to_track=['P'], scroll_margins=(1, 2), saccade=False, pad_char=' ')]
    for c in crops:
      c.set_engine(eng)
    eng.its_showtime()
    if fused:
      assert cropping.fuse_croppers(eng, crops, only_crops=only_crops)
    for c in crops:
      c.crop(None)
    for s in range(seed_steps[1]):
      eng.step_hashed(seed_steps[0], s, 1)
      for c in crops:
        c.crop(None)
    return eng, crops

  a, ca = build((0xC0FFEE, 41))
  blob = a.export_state(with_observation=with_observation)
  b, cb = build((0xBAD, 9))   # windows somewhere else entirely
  b.import_state(blob)
  if with_observation:        # the restored cropped planes are what the exporting engine last handed out
    for x, y in zip(ca, cb):
      assert torch.equal(x.crop(None).board, y.crop(None).board)
  for s in range(41, 41 + 30):
    a.step_hashed(0xC0FFEE, s, 1); b.step_hashed(0xC0FFEE, s, 1)
    for i, (x, y) in enumerate(zip(ca, cb)):
      ox, oy = x.crop(None), y.crop(None)
      assert torch.equal(ox.board, oy.board), 'cropper %d, step %d: the resumed window went its own way' % (i, s)
      for ch in ox.layers:
        assert torch.equal(ox.layers[ch], oy.layers[ch])
  # a checkpoint is of THESE croppers: another count is refused
  c, _ = build((1, 1))
  extra = cropping.FixedCropper((0, 0), 3, 3)
  extra.set_engine(c)
  with pytest.raises(ValueError):
    c.import_state(blob)
  for e in (a, b, c):
    e.close()


@pytest.mark.parametrize('kind', ['features', 'to_array', 'repaint'])
@pytest.mark.parametrize('with_observation', [True, False])
def test_resume_with_a_fused_epilogue(kind, with_observation):
  """A fused post-processor's array is written by the step kernel: right after import_state() it must show the
  RESTORED observation (refilled from the restored planes), not what this engine's last step wrote, and the steps
  after it must equal the exporting engine's."""
  import torch
  from pycolab_amd import rendering
  from pycolab_amd.engine import Engine
  t = helpers.load_template('scrolly_maze_L0')
  B = 400
  chars = [chr(c) for c in t.chars]

  def make():
    if kind == 'features':
      return rendering.ObservationToFeatureArray('P@#a')
    if kind == 'to_array':
      return rendering.ObservationToArray({c: (i, 2 * i, 255 - i) for i, c in enumerate(chars)}, dtype=np.uint8)
    return rendering.ObservationCharacterRepainter({'a': 'x', 'b': 'x', 'c': 'x'})

  def value(po, obs):
    out = po(obs)
    return out.board if kind == 'repaint' else out

  a = Engine.from_template(t, batch=B, auto_reset=True, seed=7)
  b = Engine.from_template(t, batch=B, auto_reset=True, seed=7)
  a.its_showtime(); b.its_showtime()
  pa, pb = make(), make()
  assert pa.fuse_into(a) and pb.fuse_into(b)
  a.step_hashed(0xC0FFEE, 0, 25); b.step_hashed(0xBAD, 0, 7)
  blob = a.export_state(with_observation=with_observation)
  b.import_state(blob)
  if with_observation:
    assert torch.equal(value(pb, b._result()[0]), value(pa, a._result()[0])), 'stale epilogue output after import'
  for s in range(25, 40):
    a.step_hashed(0xC0FFEE, s, 1); b.step_hashed(0xC0FFEE, s, 1)
    assert torch.equal(value(pb, b._result()[0]), value(pa, a._result()[0])), s
  a.close(); b.close()


def test_checkpoint_of_another_level_is_refused():
  """Same game, same board size, same cast -- another level (or parameter set): the header's template hash differs."""
  from tests.hip_adapter import HipAdapter
  a = HipAdapter(helpers.load_template('scrolly_maze_L0'), 128, seed=7)
  b = HipAdapter(helpers.load_template('scrolly_maze_L1'), 128, seed=7)
  a.reset(); b.reset()
  a.step_hashed(1, 0, 3)
  with pytest.raises(ValueError, match='template hash|level'):
    b.eng.import_state(a.eng.export_state())
  t = helpers.load_template('marauders')
  t2 = helpers.load_template('marauders')
  t2.param[5] = int(t2.param[5]) + 1      # a game-specific constant (param[0..3] are the engine's: RNG seed, environment offset)
  c, d = HipAdapter(t, 64, seed=1), HipAdapter(t2, 64, seed=1)
  c.reset(); d.reset()
  with pytest.raises(ValueError, match='template hash|level'):
    d.eng.import_state(c.eng.export_state())
  e = HipAdapter(t, 64, seed=2)           # another RNG seed: the continuation would draw other numbers
  e.reset()
  with pytest.raises(ValueError, match='template hash|level'):
    e.eng.import_state(c.eng.export_state())


def test_checkpoint_with_a_pass_through_cropper_attached():
  """a base ObservationCropper() (what a Story installs for every chapter by default) registers itself with the
  engine but has no device object and no state: export_state() / import_state() must carry on (a zero-length part keeps
  the trailer's attachment order), next to a real cropper whose window IS state."""
  import torch
  from pycolab_amd import cropping
  from pycolab_amd.engine import Engine
  t = helpers.load_template('better_scrolly_maze_L1')

  def build(steps, seed):
    eng = Engine.from_template(t, batch=200, auto_reset=True, seed=7)
    plain = cropping.ObservationCropper()
    real = cropping.ScrollingCropper(rows=7, cols=11, to_track=['P'], pad_char=' ', scroll_margins=(2, 3))
    plain.set_engine(eng); real.set_engine(eng)
    eng.its_showtime()
    for s in range(steps):
      eng.step_hashed(seed, s, 1)
      real.crop(None)
    return eng, plain, real
  a, pa, ra = build(33, 0xC0FFEE)
  blob = a.export_state(with_observation=True)
  b, pb, rb = build(5, 0xBAD)
  b.import_state(blob)
  assert torch.equal(ra.crop(None).board, rb.crop(None).board)
  for s in range(33, 53):
    a.step_hashed(0xC0FFEE, s, 1); b.step_hashed(0xC0FFEE, s, 1)
    assert torch.equal(ra.crop(None).board, rb.crop(None).board), s
  obs = b._result()[0]
  assert pb.crop(obs) is obs  # (the identity cropper stays the identity)
  a.close(); b.close()


def test_resume_with_a_feature_stack_fused_into_a_window():
  """ObservationToFeatureArray.fuse_into(engine, source=cropper) -- the step kernel writes the window's float32
  stack.  After import_state() the converter must hand out the stack of the RESTORED window, not the tensor the
  importing engine's own last step wrote."""
  import torch
  from pycolab_amd import cropping, rendering
  from pycolab_amd.engine import Engine
  t = helpers.load_template('better_scrolly_maze_L1')
  chars = ''.join(chr(c) for c in t.chars)

  def build(steps, seed):
    eng = Engine.from_template(t, batch=200, auto_reset=True, seed=7)
    cr = cropping.ScrollingCropper(rows=7, cols=11, to_track=['P'], pad_char=' ', scroll_margins=(2, 3))
    assert cropping.fuse_croppers(eng, [cr]) is None
    eng.its_showtime()
    conv = rendering.ObservationToFeatureArray(chars)
    assert conv.fuse_into(eng, source=cr)
    for s in range(steps):
      eng.step_hashed(seed, s, 1)
    return eng, cr, conv
  a, ca, fa = build(29, 0xC0FFEE)
  want = fa(ca.crop(None)).clone()
  blob = a.export_state(with_observation=True)
  b, cb, fb = build(7, 0xBAD)
  stale = fb(cb.crop(None)).clone()
  assert not torch.equal(stale, want)
  b.import_state(blob)
  got = fb(cb.crop(None))
  assert torch.equal(got, want), 'the converter handed out the pre-import stack'
  plain = rendering.ObservationToFeatureArray(chars)
  assert torch.equal(got, plain(cb.crop(None)))
  for s in range(29, 40):
    a.step_hashed(0xC0FFEE, s, 1); b.step_hashed(0xC0FFEE, s, 1)
    assert torch.equal(fa(ca.crop(None)), fb(cb.crop(None))), s
  a.close(); b.close()
