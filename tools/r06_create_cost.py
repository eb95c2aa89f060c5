#!/usr/bin/env python3
"""Where a small GPU test's wall time goes: engine creation, reset, steps, reads, close (HipAdapter as the tests use it) and the oracle
beside it, for a few templates.  python tools/r06_create_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import binding
from tests import helpers
from tests.hip_adapter import HipAdapter

def t():
  return time.perf_counter()

for name, B in (('scrolly_maze_L0', 2048), ('warehouse_L0', 1024), ('marauders', 768), ('walkers_room', 1024), ('scrolly_maze_L0', 2048)):
  tpl = helpers.load_template(name)
  t0 = t(); hip = HipAdapter(tpl, B); t1 = t(); hip.reset(); t2 = t()
  for i in range(20): hip.step_hashed(0x5EED, i, 1)
  hip.read('planes'); t3 = t()
  for nm in ('planes', 'reward', 'reward_set', 'discount', 'done', 'frame', 'error'): hip.read(nm)
  hip.sprites(); hip.curtains(); t4 = t()
  hip.eng.close(); t5 = t()
  o0 = t(); orc = binding.OracleEngine(tpl, B); orc.reset(); o1 = t()
  for i in range(20): orc.step_hashed(0x5EED, i, 1)
  o2 = t()
  print('%-18s B %5d  create %.3f  reset %.3f  20 steps+read %.3f  reads %.3f  close %.3f | oracle create+reset %.3f  20 steps %.3f s' % (
      name, B, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, o1 - o0, o2 - o1), flush=True)
