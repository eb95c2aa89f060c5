#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the scrolly_maze step path.

One "step" = one pass of the hot path (Engine.play for every environment of
the batch = one launch of pcx_scrolly_maze_step) over synthetic actions that
are already resident in HBM.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_worker(args):
  """TEST INFRASTRUCTURE use of the oracle: timed CPU baseline leg only."""
  template_path, batch, steps, offset = args
  from oracle import binding
  from pycolab_amd.compiler import GameTemplate
  t = GameTemplate.load(template_path)
  eng = binding.OracleEngine(t, batch)
  eng.reset()
  t0 = time.perf_counter()
  eng.step_hashed(0x5EED, 0, steps, env_offset=offset)
  return time.perf_counter() - t0


def cpu_baseline(template_path, budget_envs=1024, steps=512, max_procs=64):
  """The C oracle ("port") on this box's host cores, bounded sample (~10 s)."""
  import multiprocessing as mp
  cores = max(1, min(max_procs, len(os.sched_getaffinity(0))))
  jobs = [(template_path, budget_envs, steps, i * budget_envs) for i in range(cores)]
  with mp.get_context('fork').Pool(cores) as pool:
    times = pool.map(cpu_worker, jobs)
  wall = max(times)  # all workers step concurrently; construction is untimed
  total = cores * budget_envs * steps
  return {'value': total / wall, 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port',
          'sample': 'oracle/pcx_oracle.c, %d procs x %d envs x %d steps of the same workload '
                    '(stepping wall %.1f s)' % (cores, budget_envs, steps, wall)}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=200)
  ap.add_argument('--warmup', type=int, default=20)
  ap.add_argument('--batch', type=int, default=1 << 20, help='environments PER GPU (weak scaling)')
  ap.add_argument('--level', type=int, default=0)
  ap.add_argument('--game', default='scrolly_maze', choices=['scrolly_maze', 'warehouse', 'marauders', 'hello_world'],
                  help='scrolly_maze is the headline metric; the others are the parity configs of BASELINE.json')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  args = ap.parse_args()

  import torch
  rank = int(os.environ.get('RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  local = int(os.environ.get('LOCAL_RANK', '0'))
  if args.gpus != world and world > 1:
    raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
  torch.cuda.set_device(local)
  distributed = 'RANK' in os.environ and 'MASTER_ADDR' in os.environ  # launched by torch.distributed.run
  if distributed:
    import torch.distributed as dist
    dist.init_process_group('nccl', device_id=torch.device('cuda', local))

  from pycolab_amd import _native as N
  from pycolab_amd.compiler import GameTemplate
  from pycolab_amd.engine import Engine

  fixture = {'scrolly_maze': 'scrolly_maze_L%d' % args.level, 'warehouse': 'warehouse_L%d' % args.level,
             'marauders': 'marauders', 'hello_world': 'hello_world'}[args.game]
  template_path = os.path.join(ROOT, 'tests', 'golden', 'templates', fixture + '.npz')
  template = GameTemplate.load(template_path)
  B = args.batch
  eng = Engine.from_template(template, batch=B, device=local, auto_reset=True, seed=0x5EED, env_offset=rank * B)
  eng.its_showtime()

  # Synthetic action tape, resident in HBM before timing: uniform ordinary
  # actions {0..n_actions-1}, one int32 row per step.
  g = torch.Generator(device='cuda')
  g.manual_seed(0x5EED + rank)
  total = args.warmup + args.steps
  tape = torch.randint(0, template.n_actions, (total, B), dtype=torch.int32, device='cuda', generator=g)

  def barrier():
    if distributed:
      dist.barrier()
    torch.cuda.synchronize()

  for t in range(args.warmup):
    eng.step(tape[t])
  barrier()
  ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter()
  ev0.record()
  for t in range(args.warmup, total):
    eng.step(tape[t])
  ev1.record()
  barrier()
  wall = time.perf_counter() - t0
  kernel_ms = ev0.elapsed_time(ev1) / args.steps  # avg launch duration on the launch stream

  if distributed:
    w = torch.tensor([wall], dtype=torch.float64, device='cuda')
    dist.all_reduce(w, op=dist.ReduceOp.MAX)
    wall = float(w.item())
  eng.check_errors()

  if rank == 0:
    bytes_per_step = int(N.lib().pcx_engine_bytes_per_step(eng._native))
    achieved = bytes_per_step * B / (kernel_ms * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    if os.path.exists(pmc):
      try:
        rec = json.load(open(pmc))
        if rec.get('batch') == B and rec.get('level') == args.level:
          traffic = rec['bytes_per_launch']
      except Exception:  # pylint: disable=broad-except
        pass
    line = {
        'metric': 'env-steps/sec (whole node), scrolly_maze batch=1M; bit-exact vs CPU' if args.game == 'scrolly_maze'
                  else 'env-steps/sec (whole node), %s' % fixture,
        'value': world * B * args.steps / wall,
        'unit': 'env-steps/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': wall / args.steps * 1e3,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'u8',
        'data': 'synthetic',
        'config': {'workload': 'examples/%s level %d, %d envs per GPU, uniform actions 0-%d, '
                               'auto-reset episodes, full observation (board + %d layers) every step'
                               % (fixture, args.level, B, template.n_actions - 1, len(template.chars)),
                   'batch_per_gpu': B, 'global_batch': world * B, 'parallelism': 'env-shard x%d' % world},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                     'kernel': N.lib().pcx_engine_kernel_name(eng._native).decode(),
                     'kernel_ms': kernel_ms, 'algorithmic_bytes_per_env_step': bytes_per_step},
    }
    if world == 1 and not args.no_cpu_baseline:
      line['cpu_baseline'] = cpu_baseline(template_path)
    print(json.dumps(line))
  if distributed:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
