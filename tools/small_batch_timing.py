"""Per-step cost at small batches: one Python call per step vs one C call for T steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pycolab_amd.compiler import GameTemplate
from pycolab_amd.engine import Engine
t = GameTemplate.load('tests/golden/templates/scrolly_maze_L0.npz')
for B in (256, 4096, 16384):
  eng = Engine.from_template(t, batch=B, device=0, auto_reset=True)
  eng.its_showtime()
  T = 2000
  tape = torch.randint(0, 5, (T, B), dtype=torch.int32, device='cuda')
  for i in range(50): eng.step(tape[i])
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for i in range(T): eng.step(tape[i])
  torch.cuda.synchronize(); a = (time.perf_counter() - t0) / T
  eng.step_hashed(1, 0, 50)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  eng.step_hashed(1, 50, T)
  torch.cuda.synchronize(); b = (time.perf_counter() - t0) / T
  eng.step_n(tape[:50])
  torch.cuda.synchronize(); t0 = time.perf_counter()
  eng.step_n(tape)
  torch.cuda.synchronize(); c = (time.perf_counter() - t0) / T
  print('B=%6d: python loop %.1f us/step, step_hashed(T) %.1f us/step, step_n(tape) %.1f us/step' % (B, a * 1e6, b * 1e6, c * 1e6))
