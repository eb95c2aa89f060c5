#!/usr/bin/env python3
"""The stand-alone cropper kernels under rocprofv3 (their Python-side launch cost -- five HIP calls
per crop() -- hides the kernels' own duration from event timing of a Python loop):
  rocprofv3 --kernel-trace --stats ... -- python tools/crop_profile.py
  rocprofv3 --kernel-trace --pmc FETCH_SIZE ... / --pmc WRITE_SIZE ...
better_scrolly_maze L0 (45x89 board, 8 planes), 65,536 environments, the three windows of
profiles/r02_post_kernels.md, 100 crops each after fresh steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pycolab_amd import cropping
from pycolab_amd.compiler import GameTemplate
from pycolab_amd.engine import Engine
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t = GameTemplate.load(os.path.join(ROOT, 'tests', 'golden', 'templates', 'better_scrolly_maze_L0.npz'))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
eng = Engine.from_template(t, batch=B, auto_reset=True, seed=1)
obs = eng.its_showtime()[0]
eng.step_hashed(7, 0, 20)
crs = [cropping.ScrollingCropper(10, 30, ['P'], initial_offset=(0, -4)),
       cropping.ScrollingCropper(7, 10, ['P'], pad_char='#', scroll_margins=(None, 3)),
       cropping.FixedCropper((15, 34), 12, 20)]
for cr in crs:
  cr.set_engine(eng)
for i in range(100):
  eng.step_hashed(7, 20 + i, 1)
  for cr in crs:
    cr.crop(obs)
torch.cuda.synchronize()
print('done: windows 10x30 (grid %d), 7x10 (grid %d), 12x20 (grid %d) output dwords x envs' % (75 * B, 18 * B, 60 * B))
