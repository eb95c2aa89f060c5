#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the scrolly_maze step path.

One "step" = one pass of the hot path (Engine.play for every environment of
the batch = one launch of the game's fused step kernel) over synthetic actions
that are already resident in HBM.  Prints ONE JSON line on rank 0.

  python bench.py                       1 GPU, scrolly_maze L0, 1,048,576 envs (BASELINE metric config)
  python bench.py --gpus 8              8 ranks, one per GPU (spawned here through torch.distributed.run
                                        when not already launched by it); weak scaling: 1,048,576 envs per GPU
  python bench.py --gpus 8 --scaling strong   fixed global batch 1,048,576 = 131,072 envs per GPU
  python bench.py --gpus 8 --gather     also times the steps followed by the RCCL all-gather of the packed
                                        reward/discount/reward_set/done record (10 B/env), reported separately
  python bench.py --game marauders --batch 32768    the other BASELINE configs (3: marauders 32,768;
  python bench.py --game warehouse --batch 262144    4: warehouse 262,144; 2: scrolly_maze --batch 4096)
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

FIXTURES = {'scrolly_maze': 'scrolly_maze_L%d', 'warehouse': 'warehouse_L%d', 'marauders': 'marauders',
            'hello_world': 'hello_world', 'better_scrolly_maze': 'better_scrolly_maze_L%d'}


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
                    '(stepping wall %.1f s); the imported Python reference itself measured 7.6 k env-steps/s '
                    'per core in the build container (BASELINE.md) and is absent on the GPU box'
                    % (cores, budget_envs, steps, wall)}


def pmc_traffic(game, level, batch):
  """HBM bytes per launch from the committed PMC passes (profiles/hbm_traffic.json:
  WRITE_SIZE + corrected FETCH_SIZE, collected as MI355X_MICROARCH.md prescribes),
  or None when no record matches this workload."""
  path = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
  try:
    for rec in json.load(open(path)).get('records', []):
      if rec.get('game') == game and rec.get('level') == level and rec.get('batch') == batch:
        return rec['bytes_per_launch']
  except Exception:  # pylint: disable=broad-except
    pass
  return None


def spawn_ranks(n, argv):
  """`--gpus N` without a launcher: become `torch.distributed.run` with N ranks."""
  import torch
  have = torch.cuda.device_count()
  if have < n:
    raise SystemExit('bench.py: --gpus %d but this node exposes %d GPU(s)' % (n, have))
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
         '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv
  os.execv(sys.executable, cmd)


def time_steps(eng, tape, lo, hi, barrier, after_step=None):
  """Times steps [lo, hi) of the tape: (wall seconds, avg ms per step on the launch stream)."""
  import torch
  barrier()
  ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter()
  ev0.record()
  for t in range(lo, hi):
    eng.step(tape[t])
    if after_step is not None:
      after_step()
  ev1.record()
  barrier()
  wall = time.perf_counter() - t0
  return wall, ev0.elapsed_time(ev1) / (hi - lo)


def measure_config(game, level, batch, steps, warmup, device):
  """One of the other BASELINE configs on this GPU (reported inside the headline line)."""
  import torch
  from pycolab_amd import _native as N
  from pycolab_amd.compiler import GameTemplate
  from pycolab_amd.engine import Engine
  fixture = FIXTURES[game] % level if '%' in FIXTURES[game] else FIXTURES[game]
  template = GameTemplate.load(os.path.join(ROOT, 'tests', 'golden', 'templates', fixture + '.npz'))
  eng = Engine.from_template(template, batch=batch, device=device, auto_reset=True, seed=0x5EED)
  eng.its_showtime()
  g = torch.Generator(device='cuda')
  g.manual_seed(0x5EED)
  tape = torch.randint(0, template.n_actions, (warmup + steps, batch), dtype=torch.int32, device='cuda', generator=g)
  for t in range(warmup):
    eng.step(tape[t])
  _, kernel_ms = time_steps(eng, tape, warmup, warmup + steps, torch.cuda.synchronize)
  eng.check_errors()
  bps = int(N.lib().pcx_engine_bytes_per_step(eng._native))
  out = {'workload': 'examples/%s, %d envs' % (fixture, batch), 'ms_per_step': kernel_ms,
         'env_steps_per_s': batch / (kernel_ms * 1e-3),
         'kernel': N.lib().pcx_engine_kernel_name(eng._native).decode(), 'algorithmic_bytes_per_env_step': bps,
         'hbm_frac': bps * batch / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
         'traffic': pmc_traffic(game, level, batch)}
  eng.close()
  return out


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=200)
  ap.add_argument('--warmup', type=int, default=20)
  ap.add_argument('--batch', type=int, default=1 << 20,
                  help='environments per GPU (--scaling weak) or in total (--scaling strong)')
  ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'])
  ap.add_argument('--gather', action='store_true',
                  help='also time steps followed by the all-gather of the packed 10 B/env step results')
  ap.add_argument('--level', type=int, default=0)
  ap.add_argument('--game', default='scrolly_maze', choices=sorted(FIXTURES),
                  help='scrolly_maze is the headline metric; the others are the parity configs of BASELINE.json')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-other-configs', action='store_true',
                  help='skip the short measurements of BASELINE configs 2-4 added to the N=1 headline line')
  args = ap.parse_args()

  launched = 'RANK' in os.environ and 'MASTER_ADDR' in os.environ  # by torch.distributed.run
  if args.gpus > 1 and not launched:
    spawn_ranks(args.gpus, sys.argv[1:])  # does not return
  rank = int(os.environ.get('RANK', '0')) if launched else 0
  world = int(os.environ.get('WORLD_SIZE', '1')) if launched else 1
  local = int(os.environ.get('LOCAL_RANK', '0')) if launched else 0
  if args.gpus != world:
    raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))

  import torch
  if torch.cuda.device_count() <= local:
    raise SystemExit('bench.py: rank %d has no GPU (device_count=%d)' % (rank, torch.cuda.device_count()))
  torch.cuda.set_device(local)
  distributed = world > 1 or launched
  if distributed:
    import torch.distributed as dist
    dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    assert dist.get_world_size() == world

  from pycolab_amd import _native as N
  from pycolab_amd import distributed as pdist
  from pycolab_amd.compiler import GameTemplate
  from pycolab_amd.engine import Engine

  fixture = FIXTURES[args.game] % args.level if '%' in FIXTURES[args.game] else FIXTURES[args.game]
  template_path = os.path.join(ROOT, 'tests', 'golden', 'templates', fixture + '.npz')
  template = GameTemplate.load(template_path)
  if args.scaling == 'strong':
    lo, hi = pdist.shard_range(args.batch, rank, world)
    B, global_batch = hi - lo, args.batch
  else:
    B, global_batch, lo = args.batch, args.batch * world, rank * args.batch
  eng = Engine.from_template(template, batch=B, device=local, auto_reset=True, seed=0x5EED, env_offset=lo)
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

  def max_over_ranks(x):
    if not distributed:
      return x
    w = torch.tensor([x], dtype=torch.float64, device='cuda')
    dist.all_reduce(w, op=dist.ReduceOp.MAX)
    return float(w.item())

  for t in range(args.warmup):
    eng.step(tape[t])
  wall, kernel_ms = time_steps(eng, tape, args.warmup, total, barrier)
  wall = max_over_ranks(wall)
  eng.check_errors()

  gather = None
  if args.gather and distributed:
    sg = pdist.ScalarGather(eng.scalars_packed, global_batch=global_batch)
    sg.gather()  # communicator warm-up
    gwall, _ = time_steps(eng, tape, args.warmup, total, barrier, after_step=sg.gather)
    gwall = max_over_ranks(gwall)
    gather = {'collective': 'all_gather_into_tensor over RCCL, one per step, 10 B/env packed record, no host sync',
              'bytes_per_rank_per_step': 10 * B, 'ms_per_step_with_gather': gwall / args.steps * 1e3,
              'value_with_gather': global_batch * args.steps / gwall}

  if rank == 0:
    bytes_per_step = int(N.lib().pcx_engine_bytes_per_step(eng._native))
    achieved = bytes_per_step * B / (kernel_ms * 1e-3) / 1e9
    traffic = pmc_traffic(args.game, args.level, B)
    line = {
        'metric': 'env-steps/sec (whole node), scrolly_maze batch=1M; bit-exact vs CPU' if args.game == 'scrolly_maze'
                  else 'env-steps/sec (whole node), %s' % fixture,
        'value': global_batch * args.steps / wall,
        'unit': 'env-steps/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': wall / args.steps * 1e3,
        'higher_is_better': True,
        'scaling': args.scaling,
        'vs_baseline': None,
        'dtype': 'u8',
        'data': 'synthetic',
        'config': {'workload': 'examples/%s level %d, %d envs per GPU, uniform actions 0-%d, '
                               'auto-reset episodes, full observation (board + %d layers) every step'
                               % (fixture, args.level, B, template.n_actions - 1, len(template.chars)),
                   'batch_per_gpu': B, 'global_batch': global_batch, 'parallelism': 'env-shard x%d' % world},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                     'kernel': N.lib().pcx_engine_kernel_name(eng._native).decode(),
                     'kernel_ms': kernel_ms, 'algorithmic_bytes_per_env_step': bytes_per_step},
    }
    if gather is not None:
      line['gather'] = gather
    del tape
    eng.close()
    if world == 1 and args.game == 'scrolly_maze' and not args.no_other_configs:
      # BASELINE configs 2-4 on the same GPU, same run (their own kernels and rooflines), then the other two
      # hand-written kernels: SURVEY 8 f-1 (better_scrolly_maze, 45x89 board) and config 1's game on the GPU
      line['other_configs'] = [measure_config('scrolly_maze', 0, 4096, 200, 20, local),
                               measure_config('marauders', 0, 32768, 200, 20, local),
                               measure_config('warehouse', 0, 262144, 100, 10, local),
                               measure_config('better_scrolly_maze', 0, 65536, 50, 10, local),
                               measure_config('hello_world', 0, 1048576, 50, 10, local)]
    if world == 1 and not args.no_cpu_baseline:
      line['cpu_baseline'] = cpu_baseline(template_path)
    print(json.dumps(line))
  if distributed:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
