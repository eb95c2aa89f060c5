#!/usr/bin/env python3
"""Same-box A/B of pcx_warehouse_step's launch shapes (round 5): persistent workers (PCX_WM_WORKERS / _PER_CU / _LOCK /
_DYNAMIC) against the round-2 shape (PCX_WM_PW=0), one engine per batch, variants interleaved, kernel ms by HIP events.
  python tools/wm_sweep.py --game warehouse --batches 262144,1048576"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PREFIX = {'warehouse': 'PCX_WM_', 'hello_world': 'PCX_HW_', 'marauders': 'PCX_EM_', 'better_scrolly_maze': 'PCX_BS_'}
FIXTURE = {'warehouse': 'warehouse_L0', 'hello_world': 'hello_world', 'marauders': 'marauders', 'better_scrolly_maze': 'better_scrolly_maze_L0'}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--game', default='warehouse')
  ap.add_argument('--batches', default='262144,1048576')
  ap.add_argument('--steps', type=int, default=100)
  ap.add_argument('--repeats', type=int, default=3)
  ap.add_argument('--variants', default='pw0,w6k2,w4k2,w8k2,w8k3,w6k3,w4k1,w3k1,w6k2x2,w4k2x2,w3k1x2,w2k1x3,w6k2d,w6k2s')
  args = ap.parse_args()
  import torch
  from pycolab_amd import _native as N
  from pycolab_amd.compiler import GameTemplate
  from pycolab_amd.engine import Engine
  pre = PREFIX[args.game]

  def knobs(v):
    if v == 'pw0':
      return {pre + 'PW': 0}
    if v == 'auto':
      return {}
    import re
    m = re.match(r'w(\d+)k(\d+)(?:x(\d+))?([ds])?$', v)
    kw = {pre + 'WORKERS': m.group(1), pre + 'LOCK': m.group(2), pre + 'PER_CU': m.group(3) or 1}
    if m.group(4):
      kw[pre + 'DYNAMIC'] = 1 if m.group(4) == 'd' else 0
    return kw
  template = GameTemplate.load(os.path.join(ROOT, 'tests', 'golden', 'templates', FIXTURE[args.game] + '.npz'))
  names = args.variants.split(',')
  for B in [int(x) for x in args.batches.split(',')]:
    for k in list(os.environ):
      if k.startswith(pre):
        del os.environ[k]
    eng = Engine.from_template(template, batch=B, device=0, auto_reset=True, seed=0x5EED)
    eng.its_showtime()
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    tape = torch.randint(0, template.n_actions, (10 + args.steps, B), dtype=torch.int32, device='cuda', generator=g)
    bps = int(N.lib().pcx_engine_bytes_per_step(eng._native))
    times = {v: [] for v in names}
    shapes = {}
    for rep in range(args.repeats):
      for v in names:
        for k in list(os.environ):
          if k.startswith(pre):
            del os.environ[k]
        for k, x in knobs(v).items():
          os.environ[k] = str(x)
        for t in range(10):
          eng.step(tape[t])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(10, 10 + args.steps):
          eng.step(tape[t])
        e1.record(); torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / args.steps)
        shapes[v] = int(N.lib().pcx_engine_launch_shape(eng._native))
    eng.check_errors()
    for v in names:
      xs = sorted(times[v]); med = xs[len(xs) // 2]
      print('%8d  %-10s shape %2d  min %.4f  median %.4f ms   %.3f of 8 TB/s' % (B, v, shapes[v], xs[0], med, bps * B / (med * 1e-3) / 8e12), flush=True)
    eng.close()


if __name__ == '__main__':
  main()
