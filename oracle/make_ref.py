#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: builds oracle/_ref/ -- the reference itself, ready to be imported where /root/reference is not.

The reference (google-deepmind/pycolab) is pure Python, so "building" it means compiling its modules, from the
sources where they lie under /root/reference, to sourceless bytecode: oracle/_ref/pycolab/**/<module>.pyc (the
interpreter imports a .pyc that sits where the .py would).  Nothing of the reference's source text is copied;
oracle/_ref/ is git-ignored (it stays out of history) and NOT gpurun-ignored (it travels to the GPU box with the other
build products, where /root/reference does not exist).  `__graft_entry__.build()` runs this when the reference is
present; on the GPU box the prebuilt files are used as they are.

Who may import oracle/_ref: tests/ (the live differential tests), bench.py's cpu_baseline leg (the reference timed on
the host cores) -- never anything under pycolab_amd/.

  python oracle/make_ref.py [--reference /root/reference] [--check]
"""
import argparse
import os
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_ref')
SKIP_DIRS = ('__pycache__',)
# of the reference's tests/ only the test ENTITIES (TestMazeWalker, TestScrolly, ...: what the prefab-only scenarios and the
# random walker games are built from); the test cases themselves are harvested into tests/golden/reftests by
# harvest_reference_tests.py
TESTS_KEPT = ('__init__.py', 'test_things.py')


def build(reference):
  src_root = os.path.join(reference, 'pycolab')
  if not os.path.isdir(src_root):
    raise SystemExit('make_ref: %s has no pycolab package' % reference)
  if os.path.isdir(OUT):
    shutil.rmtree(OUT)
  n = 0
  for dirpath, dirnames, filenames in os.walk(src_root):
    dirnames[:] = sorted(d for d in dirnames if d not in SKIP_DIRS)
    rel = os.path.relpath(dirpath, reference)
    for name in sorted(filenames):
      if not name.endswith('.py') or (os.path.basename(dirpath) == 'tests' and name not in TESTS_KEPT):
        continue
      dst = os.path.join(OUT, rel, name + 'c')
      os.makedirs(os.path.dirname(dst), exist_ok=True)
      # dfile: the path tracebacks name; UNCHECKED_HASH: valid without the source next to it
      py_compile.compile(os.path.join(dirpath, name), cfile=dst, dfile='<reference>/' + os.path.join(rel, name), doraise=True,
                         invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
      n += 1
  with open(os.path.join(OUT, 'BUILT_BY'), 'w') as f:
    f.write('oracle/make_ref.py, Python %d.%d, %d modules\n' % (sys.version_info[0], sys.version_info[1], n))
  return n


def available():
  """The directory to put on sys.path to import the reference: /root/reference (or $PCX_REFERENCE) where it exists,
  else the prebuilt oracle/_ref, else None."""
  for path in (os.environ.get('PCX_REFERENCE'), '/root/reference', OUT):
    if path and (os.path.isfile(os.path.join(path, 'pycolab', 'engine.py')) or os.path.isfile(os.path.join(path, 'pycolab', 'engine.pyc'))):
      return path
  return None


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--reference', default=os.environ.get('PCX_REFERENCE', '/root/reference'))
  ap.add_argument('--check', action='store_true', help='import the built package in a fresh interpreter')
  args = ap.parse_args()
  n = build(args.reference)
  print('oracle/_ref: %d modules compiled from %s' % (n, args.reference))
  if args.check:
    import subprocess
    code = ('import sys; sys.path.insert(0, %r); import pycolab.examples.scrolly_maze as m; g = m.make_game(0); '
            'o, r, d = g.its_showtime(); print(o.board.shape, m.__file__)' % OUT)
    subprocess.check_call([sys.executable, '-c', code], cwd='/')


if __name__ == '__main__':
  main()
