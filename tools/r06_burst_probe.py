#!/usr/bin/env python3
"""Why `bench.py --steps 20 --warmup 5` reports ~1.5 % more per step than 200-step repeats: per-launch durations (HIP events
around every launch) of 20-step bursts that start on an idle GPU (barrier + synchronize before, as the bench contract asks)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from pycolab_amd.compiler import GameTemplate  # noqa: E402
from pycolab_amd.engine import Engine  # noqa: E402
import bench  # noqa: E402

B = 1 << 20
t = GameTemplate.load(os.path.join(ROOT, 'tests', 'golden', 'templates', 'scrolly_maze_L0.npz'))
eng = Engine.from_template(t, batch=B, device=0, auto_reset=True, seed=0x5EED)
eng.its_showtime()
g = torch.Generator(device='cuda'); g.manual_seed(1)
tape = torch.randint(0, t.n_actions, (64, B), dtype=torch.int32, device='cuda', generator=g)
for i in range(40):
  eng.step(tape[i % 64])
bench.settle_tuner(eng, lambda k: tape[k % 64], 0)
for idle_ms in (0, 1, 20, 200):
  rows = []
  for rep in range(5):
    torch.cuda.synchronize()
    time.sleep(idle_ms * 1e-3)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    evs[0].record()
    for k in range(20):
      eng.step(tape[k])
      evs[k + 1].record()
    torch.cuda.synchronize()
    rows.append([evs[k].elapsed_time(evs[k + 1]) for k in range(20)])
  med = [sorted(r[k] for r in rows)[2] for k in range(20)]
  print('idle %3d ms before the burst: launches 1..20 (median of 5 bursts, ms): %s  mean %.4f' % (
      idle_ms, ' '.join('%.3f' % v for v in med), sum(med) / 20))
