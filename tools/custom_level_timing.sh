#!/bin/bash
# Unshipped levels: the hand-written kernels' run-time-shape instances against the table-driven kernel
# (PCX_FORCE_GENERIC=1), HIP events, same box.
for G in 0 1; do PCX_FORCE_GENERIC=$G python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from pycolab_amd import _native as N
from pycolab_amd.compiler import GameTemplate
from pycolab_amd.engine import Engine
def timed(fn, steps=100):
  for _ in range(10): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(steps): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / steps
for name, batch in (('warehouse_custom_C', 262144), ('warehouse_custom_D', 262144), ('better_scrolly_custom_A', 131072),
                    ('better_scrolly_custom_B', 262144), ('warehouse_L0', 262144), ('better_scrolly_maze_L1', 131072)):
  t = GameTemplate.load('tests/golden/templates/%s.npz' % name)
  eng = Engine.from_template(t, batch=batch, auto_reset=True, seed=1)
  eng.its_showtime()
  tape = torch.randint(0, t.n_actions, (16, batch), dtype=torch.int32, device='cuda')
  c = [0]
  def one():
    eng.step(tape[c[0] % 16]); c[0] += 1
  ms = timed(one)
  bps = int(N.lib().pcx_engine_bytes_per_step(eng._native))
  print('%-26s %8d envs  %-24s %.4f ms  %.1f %% of 8 TB/s' % (name, batch, N.lib().pcx_engine_kernel_name(eng._native).decode(), ms,
                                                          100 * bps * batch / (ms * 1e-3) / 8e12))
  eng.close()
PY
done
