#!/usr/bin/env python3
"""better_scrolly_maze L0, 65,536 envs, the example's three croppers fused, windows only: 30 steps (for a PMC pass:
rocprofv3 --kernel-trace --pmc WRITE_SIZE / FETCH_SIZE -- python tools/fused_crop_pmc.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pycolab_amd import cropping
from pycolab_amd.compiler import GameTemplate
from pycolab_amd.engine import Engine
t = GameTemplate.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/templates/better_scrolly_maze_L0.npz'))
eng = Engine.from_template(t, batch=65536, auto_reset=True, seed=1)
crs = [cropping.ScrollingCropper(10, 30, ['P'], initial_offset=(-2, -12)),
       cropping.ScrollingCropper(7, 10, ['c'], pad_char=' ', scroll_margins=(None, 3)),
       cropping.FixedCropper((3, 9), 12, 20, pad_char=' ')]
cropping.fuse_croppers(eng, crs, only_crops=(len(sys.argv) < 2 or sys.argv[1] != 'full'))
eng.its_showtime()
tape = torch.randint(0, 5, (30, eng.batch), dtype=torch.int32, device='cuda')
for i in range(30):
  eng.step(tape[i])
torch.cuda.synchronize()
print('done')
