#!/usr/bin/env python3
"""Same-box, same-process A/B of the launch shapes of pcx_scrolly_maze_step (round 4).

One engine per batch size; the launch shape is chosen by environment knobs that ScrollyMazeBackend::launch
reads at every launch, so the variants alternate on the same state, the same tape and the same box:
  python tools/ps_sweep.py --batches 131072,262144,1048576 --steps 100 --repeats 3 --out gpurun_out/ps_sweep.json
Prints, per batch and variant, the kernel ms per step (HIP events on the launch stream; min / median over the
repeats) and the fraction of 8 TB/s its algorithmic bytes come to."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KNOBS = ('PCX_SM_STEAL', 'PCX_SM_BAKED', 'PCX_SM_TAIL', 'PCX_SM_TAIL_UNIT', 'PCX_DEBUG', 'PCX_SM_WAVES', 'PCX_SM_LOCK', 'PCX_SM_SHAPE', 'PCX_SM_UNIT', 'PCX_SM_DYNAMIC', 'PCX_SM_PER_CU', 'PCX_WAVES_PER_CU',
         'PCX_SM_CODES', 'PCX_SM_GRID')

VARIANTS = {
    'auto':        {},
    'unbaked':     {'PCX_SM_BAKED': 0},
    'auto_t0':     {'PCX_SM_TAIL': 0},
    'auto_t1':     {'PCX_SM_TAIL': 1},
    'auto_t3':     {'PCX_SM_TAIL': 3},
    'auto_t2_u32': {'PCX_SM_TAIL': 2, 'PCX_SM_TAIL_UNIT': 32},
    'auto_t2_u8':  {'PCX_SM_TAIL': 2, 'PCX_SM_TAIL_UNIT': 8},
    'auto_dyn':    {'PCX_SM_DYNAMIC': 1},
    'auto_dyn_t0': {'PCX_SM_DYNAMIC': 1, 'PCX_SM_TAIL': 0},
    'auto_static': {'PCX_SM_DYNAMIC': 0},
    'head':        {'PCX_SM_SHAPE': 0},
    'head_w7':     {'PCX_SM_SHAPE': 0, 'PCX_WAVES_PER_CU': 7},
    'C':           {'PCX_SM_SHAPE': 3},
    'C_nolock':    {'PCX_SM_SHAPE': 3, 'PCX_SM_LOCK': 0},
    'C_static':    {'PCX_SM_SHAPE': 3, 'PCX_SM_DYNAMIC': 0},
    'C_cu2':       {'PCX_SM_SHAPE': 3, 'PCX_SM_PER_CU': 2},
    'C_w6x1_k3':   {'PCX_SM_SHAPE': 3, 'PCX_SM_WAVES': 6, 'PCX_SM_PER_CU': 1, 'PCX_SM_LOCK': 3},
    'C_w6x1_k2':   {'PCX_SM_SHAPE': 3, 'PCX_SM_WAVES': 6, 'PCX_SM_PER_CU': 1, 'PCX_SM_LOCK': 2},
    'C2x3s':       {'PCX_SM_SHAPE': 3, 'PCX_SM_WAVES': 2, 'PCX_SM_PER_CU': 3, 'PCX_SM_LOCK': 1, 'PCX_SM_DYNAMIC': 0},
    'C2x4s':       {'PCX_SM_SHAPE': 3, 'PCX_SM_WAVES': 2, 'PCX_SM_PER_CU': 4, 'PCX_SM_LOCK': 1, 'PCX_SM_DYNAMIC': 0},
    'C4x2_k2s':    {'PCX_SM_SHAPE': 3, 'PCX_SM_WAVES': 4, 'PCX_SM_PER_CU': 2, 'PCX_SM_LOCK': 2, 'PCX_SM_DYNAMIC': 0},
    'C3x3_k2':     {'PCX_SM_SHAPE': 3, 'PCX_SM_WAVES': 3, 'PCX_SM_PER_CU': 3, 'PCX_SM_LOCK': 2},
    'C2x5':        {'PCX_SM_SHAPE': 3, 'PCX_SM_WAVES': 2, 'PCX_SM_PER_CU': 5, 'PCX_SM_LOCK': 1},
    'C2x4':        {'PCX_SM_SHAPE': 3, 'PCX_SM_WAVES': 2, 'PCX_SM_PER_CU': 4, 'PCX_SM_LOCK': 1},
    'C2x3':        {'PCX_SM_SHAPE': 3, 'PCX_SM_WAVES': 2, 'PCX_SM_PER_CU': 3, 'PCX_SM_LOCK': 1},
    'C3x3':        {'PCX_SM_SHAPE': 3, 'PCX_SM_WAVES': 3, 'PCX_SM_PER_CU': 3, 'PCX_SM_LOCK': 1},
    'C3x3s':       {'PCX_SM_SHAPE': 3, 'PCX_SM_WAVES': 3, 'PCX_SM_PER_CU': 3, 'PCX_SM_LOCK': 1, 'PCX_SM_DYNAMIC': 0},
    'C3x4':        {'PCX_SM_SHAPE': 3, 'PCX_SM_WAVES': 3, 'PCX_SM_PER_CU': 4, 'PCX_SM_LOCK': 1},
    'C4x2_k2':     {'PCX_SM_SHAPE': 3, 'PCX_SM_WAVES': 4, 'PCX_SM_PER_CU': 2, 'PCX_SM_LOCK': 2},
    'C4x2_k1':     {'PCX_SM_SHAPE': 3, 'PCX_SM_WAVES': 4, 'PCX_SM_PER_CU': 2, 'PCX_SM_LOCK': 1},
    'C4x3_k1':     {'PCX_SM_SHAPE': 3, 'PCX_SM_WAVES': 4, 'PCX_SM_PER_CU': 3, 'PCX_SM_LOCK': 1},
    'C5x2_k2':     {'PCX_SM_SHAPE': 3, 'PCX_SM_WAVES': 5, 'PCX_SM_PER_CU': 2, 'PCX_SM_LOCK': 2},
    'C10x1_k3':    {'PCX_SM_SHAPE': 3, 'PCX_SM_WAVES': 10, 'PCX_SM_PER_CU': 1, 'PCX_SM_LOCK': 3},
    'C10x1_k4':    {'PCX_SM_SHAPE': 3, 'PCX_SM_WAVES': 10, 'PCX_SM_PER_CU': 1, 'PCX_SM_LOCK': 4},
    'Cs':          {'PCX_SM_SHAPE': 3, 'PCX_SM_DYNAMIC': 0},
    'Cs_nolock':   {'PCX_SM_SHAPE': 3, 'PCX_SM_DYNAMIC': 0, 'PCX_SM_LOCK': 0},
    'Cs_w3':       {'PCX_SM_SHAPE': 3, 'PCX_SM_DYNAMIC': 0, 'PCX_SM_WAVES': 3, 'PCX_SM_PER_CU': 2},
    'Cs_w4':       {'PCX_SM_SHAPE': 3, 'PCX_SM_DYNAMIC': 0, 'PCX_SM_WAVES': 4, 'PCX_SM_PER_CU': 1},
    'Cs_w6':       {'PCX_SM_SHAPE': 3, 'PCX_SM_DYNAMIC': 0, 'PCX_SM_WAVES': 6, 'PCX_SM_PER_CU': 1},
    'Cs_w1_cu6':   {'PCX_SM_SHAPE': 3, 'PCX_SM_DYNAMIC': 0, 'PCX_SM_WAVES': 1, 'PCX_SM_PER_CU': 6},
}


def set_knobs(kw):
  for k in KNOBS:
    os.environ.pop(k, None)
  for k, v in kw.items():
    os.environ[k] = str(v)


def profile(eng, actions, name, B):
  import ctypes
  import numpy as np
  import torch
  from pycolab_amd import _native as N
  os.environ['PCX_SM_PROF'] = '1'
  eng.step(actions)
  torch.cuda.synchronize()
  N.check(N.lib().pcx_engine_debug_counters(eng._native, None, -1))  # (clears the timers)
  eng.step(actions)
  torch.cuda.synchronize()
  os.environ.pop('PCX_SM_PROF', None)
  buf = np.zeros(16 * 65536, np.uint32)
  N.check(N.lib().pcx_engine_debug_counters(eng._native, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), buf.size))
  p = buf.reshape(-1, 16).astype(np.float64) * 0.01  # us
  live = buf.reshape(-1, 16)[:, 0] > 0
  n = int(live.sum())
  if not n:
    print('          (no timers)')
    return
  p = p[live]
  units = buf.reshape(-1, 16)[live, 0].astype(np.float64)
  per_unit = lambda col: float((p[:, col].sum()) / units.sum())
  if buf.reshape(-1, 16)[live, 8].any():  # logic / render pairs
    ru = buf.reshape(-1, 16)[live, 8].astype(np.float64)
    print('          [%s] %d pairs, %.1f units each | logic wave per unit: inbox wait %.1f, buffer wait %.1f, stepping %.1f, '
          'ticket wait %.1f us; lifetime %.0f us | render wave per unit: wait %.1f, streaming %.1f us; lifetime %.0f us'
          % (name, n, units.mean(), per_unit(1), per_unit(2), per_unit(3), per_unit(4), p[:, 5].mean(),
             float(p[:, 9].sum() / ru.sum()), float(p[:, 10].sum() / ru.sum()), p[:, 11].mean()), flush=True)
  else:
    print('          [%s] %d workers, %.1f units each (min %d max %d) | per unit: inbox wait %.1f, stepping %.1f, mutex wait %.1f, streaming %.1f us; '
          'lifetime mean %.0f min %.0f max %.0f us' % (name, n, units.mean(), units.min(), units.max(), per_unit(1), per_unit(3), per_unit(2),
                                                       per_unit(10), p[:, 5].mean(), p[:, 5].min(), p[:, 5].max()), flush=True)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--batches', default='131072,262144,1048576')
  ap.add_argument('--variants', default=','.join(VARIANTS))
  ap.add_argument('--extra', default='', help='more variants: name:K=V+K=V,name2:...')
  ap.add_argument('--steps', type=int, default=100)
  ap.add_argument('--warmup', type=int, default=10)
  ap.add_argument('--repeats', type=int, default=3)
  ap.add_argument('--level', type=int, default=0)
  ap.add_argument('--out', default=None)
  ap.add_argument('--prof', action='store_true', help='after timing a persistent variant: one launch with the phase '
                  'timers on (PCX_SM_PROF), summarised per workgroup (mean us per unit and per launch)')
  args = ap.parse_args()
  import torch
  from pycolab_amd import _native as N
  from pycolab_amd.compiler import GameTemplate
  from pycolab_amd.engine import Engine
  variants = {k: VARIANTS[k] for k in args.variants.split(',') if k in VARIANTS}
  for spec in [x for x in args.extra.split(',') if x]:
    name, kv = spec.split(':')
    variants[name] = dict(p.split('=') for p in kv.split('+'))
  template = GameTemplate.load(os.path.join(ROOT, 'tests', 'golden', 'templates', 'scrolly_maze_L%d.npz' % args.level))
  results = []
  for B in [int(x) for x in args.batches.split(',')]:
    set_knobs({})
    eng = Engine.from_template(template, batch=B, device=0, auto_reset=True, seed=0x5EED)
    eng.its_showtime()
    g = torch.Generator(device='cuda')
    g.manual_seed(0x5EED)
    K, W = args.steps, args.warmup
    tape = torch.randint(0, template.n_actions, (W + K, B), dtype=torch.int32, device='cuda', generator=g)
    bps = int(N.lib().pcx_engine_bytes_per_step(eng._native))
    times = {k: [] for k in variants}
    shapes = {}
    for rep in range(args.repeats):
      for name, kw in list(variants.items()):
        set_knobs(kw)
        try:
          eng.step(tape[0])
        except Exception as exc:  # pylint: disable=broad-except
          print('          [%s] cannot launch: %s' % (name, exc), flush=True)
          del variants[name]
          continue
        for t in range(W):
          eng.step(tape[t])
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for t in range(W, W + K):
          eng.step(tape[t])
        ev1.record()
        torch.cuda.synchronize()
        times[name].append(ev0.elapsed_time(ev1) / K)
        shapes[name] = int(N.lib().pcx_engine_launch_shape(eng._native))
        if args.prof and rep == 0 and shapes[name] in (1, 2, 3, 5):
          profile(eng, tape[W], name, B)
    eng.check_errors()
    for name in variants:
      xs = sorted(times[name])
      med = xs[len(xs) // 2]
      rec = {'batch': B, 'variant': name, 'knobs': variants[name], 'shape': shapes[name], 'ms_min': xs[0], 'ms_median': med,
             'ms_all': times[name], 'frac_of_8TBs_at_median': bps * B / (med * 1e-3) / 8e12}
      results.append(rec)
      print('%8d  %-12s shape %2d  min %.4f  median %.4f ms   %.3f of 8 TB/s' % (B, name, shapes[name], xs[0], med,
                                                                                 rec['frac_of_8TBs_at_median']), flush=True)
    eng.close()
    del tape
  set_knobs({})
  if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump({'steps': args.steps, 'repeats': args.repeats, 'results': results}, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
  main()
