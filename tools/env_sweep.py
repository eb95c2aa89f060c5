#!/usr/bin/env python3
"""Same-box A/B of launch knobs / ablations that the library reads from the environment at every launch: one engine per
batch, the variants interleaved, kernel ms by HIP events on the launch stream.
  python tools/env_sweep.py --game better_scrolly_maze --batches 65536 --variants "base;logic:PCX_DEBUG=2;coop:PCX_COOP_BELOW=100"
A variant is `name` or `name:KEY=VALUE,KEY=VALUE`; keys that start with `!` are read at engine creation (a fresh engine per
variant instead of interleaving on one)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FIXTURE = {'warehouse': 'warehouse_L0', 'hello_world': 'hello_world', 'marauders': 'marauders', 'better_scrolly_maze': 'better_scrolly_maze_L0',
           'scrolly_maze': 'scrolly_maze_L0'}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--game', default='warehouse')
  ap.add_argument('--fixture', default=None, help='a template of tests/golden/templates instead of the game\'s shipped level')
  ap.add_argument('--batches', default='262144')
  ap.add_argument('--steps', type=int, default=50)
  ap.add_argument('--repeats', type=int, default=3)
  ap.add_argument('--cardinal-fields', type=int, default=0)
  ap.add_argument('--variants', default='base')
  args = ap.parse_args()
  import torch
  from pycolab_amd import _native as N
  from pycolab_amd.compiler import GameTemplate
  from pycolab_amd.engine import Engine
  variants = []
  for v in args.variants.split(';'):
    name, _, kv = v.partition(':')
    variants.append((name, dict(x.split('=', 1) for x in kv.split(',') if x)))
  keys = {k.lstrip('!') for _, kw in variants for k in kw}
  fresh = any(k.startswith('!') for _, kw in variants for k in kw)
  template = GameTemplate.load(os.path.join(ROOT, 'tests', 'golden', 'templates', (args.fixture or FIXTURE[args.game]) + '.npz'))

  def set_env(kw):
    for k in keys:
      os.environ.pop(k, None)
    for k, x in kw.items():
      os.environ[k.lstrip('!')] = x

  for B in [int(x) for x in args.batches.split(',')]:
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    W = 12
    if args.cardinal_fields:
      values = torch.tensor([0, 2, 4, 6, 8], dtype=torch.int32, device='cuda')
      tape = torch.zeros((W + args.steps, B), dtype=torch.int32, device='cuda')
      for f in range(args.cardinal_fields):
        tape |= values[torch.randint(0, 5, (W + args.steps, B), device='cuda', generator=g)] << (4 * f)
    else:
      tape = torch.randint(0, template.n_actions, (W + args.steps, B), dtype=torch.int32, device='cuda', generator=g)
    times = {name: [] for name, _ in variants}
    shapes = {}
    eng = None
    for rep in range(args.repeats):
      for name, kw in variants:
        set_env(kw)
        if eng is None or fresh:
          if eng is not None:
            eng.close()
          eng = Engine.from_template(template, batch=B, device=0, auto_reset=True, seed=0x5EED)
          eng.its_showtime()
          for t in range(W + 40):
            eng.step(tape[t % W])  # (the launch-shape tuners finish on the engine's first launches)
        for t in range(W):
          eng.step(tape[t])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(W, W + args.steps):
          eng.step(tape[t])
        e1.record(); torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / args.steps)
        shapes[name] = (N.lib().pcx_engine_kernel_name(eng._native).decode(), int(N.lib().pcx_engine_launch_shape(eng._native)))
    bps = int(N.lib().pcx_engine_bytes_per_step(eng._native))
    set_env({})
    for name, _ in variants:
      xs = sorted(times[name]); med = xs[len(xs) // 2]
      print('%8d  %-14s %s shape %2d  min %.4f  median %.4f ms   %.3f of 8 TB/s' % (B, name, shapes[name][0], shapes[name][1], xs[0], med, bps * B / (med * 1e-3) / 8e12), flush=True)
    eng.close()


if __name__ == '__main__':
  main()
