#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: tests/golden/digests/<game>.npz -- the reference itself at BASELINE.md section 3's gate size.

For scrolly_maze L0, warehouse_manager L0, extraterrestrial_marauders and better_scrolly_maze L0 the imported
reference (oracle/ref_live.py) steps 4,096 environments for 256 steps, resets included, on the bench's own tape --
environments [0, 4096) and [1,044,480, 1,048,576): the head and the tail of the headline batch -- and what play()
returned is reduced to digests (ref_live.chunk_digests): per step and chunk of 256 environments the first 8 bytes
of a SHA-256 over board | (scrolly_maze_L1_unoccluded: every layer |) reward | reward_set | discount | done, and per step the SHA-256 over that step's full chunk
digests.  tests/test_gate_digests.py requires the C oracle (CPU suite) and the HIP path (GPU suite, the tail offset
inside a real 1,048,576-environment engine for scrolly_maze) to reproduce them.

About 5 minutes on 8 cores.  PCX_GOLDEN_OUT: write elsewhere (the reproducibility test regenerates a sample).
  python oracle/gen_digests.py [game ...]
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_live  # noqa: E402

N_ENVS, STEPS, SEED = 4096, 256, 0x5EED
OFFSETS = {'head': 0, 'tail': 1048576 - N_ENVS}
GAMES = ('scrolly_maze_L0', 'warehouse_L0', 'marauders', 'better_scrolly_maze_L0',
         'scrolly_maze_L1_unoccluded')  # round 6: occlusion_in_layers=False at gate size, every layer in the digest


def reduce(chunk32):
  """[frames, chunks, 32] full chunk digests -> (first 8 bytes of each, SHA-256 per frame over the frame's full digests)."""
  per_step = np.stack([np.frombuffer(hashlib.sha256(chunk32[t].tobytes()).digest(), np.uint8) for t in range(chunk32.shape[0])])
  return np.ascontiguousarray(chunk32[:, :, :8]), per_step


def main():
  out_root = os.environ.get('PCX_GOLDEN_OUT') or os.path.join(ROOT, 'tests', 'golden')
  os.makedirs(os.path.join(out_root, 'digests'), exist_ok=True)
  for name in (sys.argv[1:] or GAMES):
    arrays = {'n_envs': np.array([N_ENVS]), 'steps': np.array([STEPS]), 'seed': np.array([SEED], np.uint64)}
    for tag, off in OFFSETS.items():
      chunk32, resets = ref_live.digests(name, off, N_ENVS, STEPS, SEED)
      arrays['chunks_' + tag], arrays['steps_' + tag] = reduce(chunk32)
      arrays['offset_' + tag] = np.array([off], np.int64)
      arrays['resets_' + tag] = np.array([resets])
      print('%s %s: environments %d..%d x %d steps, %d episodes ended' % (name, tag, off, off + N_ENVS, STEPS, resets), flush=True)
    path = os.path.join(out_root, 'digests', name + '.npz')
    np.savez(path, **arrays)
    print('wrote %s (%d bytes)' % (path, os.path.getsize(path)), flush=True)


if __name__ == '__main__':
  main()
