"""Every committed golden fixture is what its generator produces from the
reference TODAY: the generators (oracle/gen_templates.py, gen_golden.py,
harvest_reference_tests.py, gen_story_golden.py, gen_ordeal_golden.py, gen_raise_golden.py; a sample of gen_digests.py) are re-run against
/root/reference into a scratch directory and every array of every file must
equal the committed one.  Runs where the reference exists (the build
container); the GPU box has no /root/reference and skips it."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import helpers

REFERENCE = os.environ.get('PCX_REFERENCE', '/root/reference')
GENERATORS = ['gen_templates.py', 'harvest_reference_tests.py', 'gen_story_golden.py', 'gen_ordeal_golden.py', 'gen_golden.py', 'gen_raise_golden.py']


@pytest.fixture(scope='module')
def regenerated(tmp_path_factory):
  if not os.path.isdir(os.path.join(REFERENCE, 'pycolab')):
    pytest.skip('the reference is not on this machine')
  out = tmp_path_factory.mktemp('regold')
  env = dict(os.environ, PCX_GOLDEN_OUT=str(out), PCX_REFERENCE=REFERENCE, PYTHONDONTWRITEBYTECODE='1')
  for gen in GENERATORS:
    r = subprocess.run([sys.executable, os.path.join(helpers.ROOT, 'oracle', gen)], env=env, cwd=helpers.ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=900)
    assert r.returncode == 0, '%s failed:\n%s' % (gen, r.stdout[-3000:])
  return str(out)


def _same_npz(a, b):
  za, zb = np.load(a), np.load(b)
  assert sorted(za.files) == sorted(zb.files), (a, sorted(za.files), sorted(zb.files))
  for k in za.files:
    assert za[k].dtype == zb[k].dtype and za[k].shape == zb[k].shape, (a, k)
    np.testing.assert_array_equal(za[k], zb[k], err_msg='%s: %s' % (os.path.basename(a), k))


@pytest.mark.parametrize('sub', ['templates', 'reftests', 'traces', 'raises'])
def test_committed_fixtures_are_what_the_generators_produce(regenerated, sub):
  committed = os.path.join(helpers.GOLDEN, sub)
  fresh = os.path.join(regenerated, sub)
  names = sorted(f for f in os.listdir(committed) if f.endswith('.npz'))
  assert names == sorted(f for f in os.listdir(fresh) if f.endswith('.npz')), 'the set of %s fixtures changed' % sub
  assert names
  for f in names:
    _same_npz(os.path.join(committed, f), os.path.join(fresh, f))
  for f in sorted(os.listdir(committed)):
    if f.endswith('.json'):
      assert open(os.path.join(committed, f)).read() == open(os.path.join(fresh, f)).read(), f


def test_shipped_fingerprints_are_what_the_generator_produces(regenerated):
  strip = lambda text: [l for l in text.splitlines() if l.startswith('    (')]
  fresh = strip(open(os.path.join(regenerated, '_shipped_fingerprints.py')).read())
  committed = strip(open(os.path.join(helpers.ROOT, 'pycolab_amd', '_shipped_fingerprints.py')).read())
  assert fresh and fresh == committed


@pytest.mark.parametrize('name', ['scrolly_maze_L0', 'warehouse_L0', 'marauders', 'better_scrolly_maze_L0', 'scrolly_maze_L1_unoccluded'])
def test_gate_digests_are_what_the_reference_produces(name):
  """tests/golden/digests (oracle/gen_digests.py: the reference at 4,096 environments x 256 steps, twice per game) take
  minutes to regenerate: a sample -- one chunk of 256 environments from the head and one from the tail, the first 96 of
  the 256 steps -- is recomputed from the live reference and must equal the committed digests."""
  from oracle import gen_digests, ref_live
  if ref_live.reference_path() is None:
    pytest.skip('the reference is not on this machine')
  fix = np.load(os.path.join(helpers.GOLDEN, 'digests', name + '.npz'))
  steps = 96
  for tag, chunk in (('head', 3), ('tail', 14)):
    off = int(fix['offset_' + tag][0])
    assert off == gen_digests.OFFSETS[tag]
    r = ref_live.run(name, off + chunk * ref_live.CHUNK, ref_live.CHUNK, steps, gen_digests.SEED)
    got = ref_live.chunk_digests(r['boards'], r['reward'], r['reward_set'], r['discount'], r['done'], r.get('layers'))[:, 0, :8]
    np.testing.assert_array_equal(got, fix['chunks_' + tag][:steps + 1, chunk], err_msg='%s %s chunk %d' % (name, tag, chunk))
