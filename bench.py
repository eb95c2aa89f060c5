#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the scrolly_maze step path.

One "step" = one pass of the hot path (Engine.play for every environment of
the batch = one launch of the game's fused step kernel) over synthetic actions
that are already resident in HBM.  Prints ONE JSON line on rank 0.

  python bench.py                       1 GPU, scrolly_maze L0, 1,048,576 envs (BASELINE metric config)
  python bench.py --gpus 8              8 ranks, one per GPU (spawned here through torch.distributed.run
                                        when not already launched by it).  BASELINE config 5 as stated: the FIXED
                                        global batch 1,048,576 sharded over the ranks (131,072 envs per GPU at
                                        N = 8; "scaling": "strong"); the weak-scaling figure (1,048,576 envs per
                                        GPU) is a second timed block of the same line ("weak_scaling"), and the
                                        steps followed by the RCCL all-gather of the packed reward / discount /
                                        reward_set / done record (10 B/env) a third ("gather")
  python bench.py --gpus 8 --scaling weak     1,048,576 envs per GPU as the headline value instead
  python bench.py --gpus 8 --no-gather  skip the all-gather block
  python bench.py --gpus 2 --oversubscribe    the same N-rank path on a node with FEWER GPUs than ranks: ranks
                                        share devices round-robin and the process group is gloo (RCCL refuses two
                                        ranks on one device); launcher, sharding, accounting and JSON are the
                                        real path's.  The line says "oversubscribed": true; its value is not a
                                        scaling measurement
  python bench.py --game marauders --batch 32768    the other BASELINE configs (3: marauders 32,768;
  python bench.py --game warehouse --batch 262144    4: warehouse 262,144; 2: scrolly_maze --batch 4096)

Timing: after W warm-up steps the run times EXACTLY K steps between barrier +
torch.cuda.synchronize() on both sides, max over ranks -- `--repeats R` (default
5) times in a row; `value` / `ms_per_step` are the MEDIAN repeat, and every
repeat is in the line ("repeats").
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


def reference_dir():
  """Where the reference can be imported from for the CPU-baseline leg: /root/reference (the build container) or the
  bytecode oracle/make_ref.py built from it under oracle/_ref (travels to the GPU box with the other build products)."""
  from oracle import make_ref
  return make_ref.available()

FIXTURES = {'scrolly_maze': 'scrolly_maze_L%d', 'warehouse': 'warehouse_L%d', 'marauders': 'marauders',
            'hello_world': 'hello_world', 'better_scrolly_maze': 'better_scrolly_maze_L%d',
            # games only the table-driven kernel steps (an unshipped marauders board; prefab MazeWalkers and Scrolly drapes in two scrolling groups)
            'marauders_custom_A': 'marauders_custom_A', 'walkers_scroll_groups': 'walkers_scroll_groups',
            # warehouse_manager level 0 stepped by the table-driven kernel (PCX_FORCE_GENERIC=1 while the engine is created): the
            # kernel's third timing fixture since round 3
            'warehouse_generic': 'warehouse_L%d',
            # round 6: a scrolly_maze level of one's own on the example's board -- pcx_scrolly_maze_step compiles its instances for it at
            # pcx_engine_create (launch_shape 7) -- next to the shipped level with the same number of coin words (level 1: launch_shape 5)
            'scrolly_custom_H': 'scrolly_custom_H',
            # round 6: the Kansas chapter of examples/ordeal.py (10x45 board, float rewards, Plot entries in the plot words) -- one
            # engine of the example's Story, stepped by pcx_generic_step
            'ordeal_kansas': 'ordeal_kansas'}


def cpu_worker(args):
  """TEST INFRASTRUCTURE use of the oracle: timed CPU baseline leg only."""
  template_path, batch, steps, offset = args
  os.environ['PCX_ORACLE_THREADS'] = '1'  # (one process per core here: the oracle's own threads over environments stay off)
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


def cpu_reference_python(game, level, seconds=10.0, max_procs=64):
  """The REAL reference (google-deepmind/pycolab, imported: engine.py:520-639 around the unchanged example game) on this
  box's host cores, same run, same workload (hashed actions, rebuild on game over): one Engine per core for `seconds`.
  TEST INFRASTRUCTURE leg (oracle/ref_timing.py); None where the reference is not available."""
  ref = reference_dir()
  if ref is None:
    return None
  from oracle import ref_timing
  return ref_timing.measure(ref, game, level, seconds=seconds, max_procs=max_procs)


def pmc_traffic(game, level, batch, kernel, launch_shape, api='step'):
  """HBM bytes per launch from the committed PMC passes (profiles/hbm_traffic.json:
  WRITE_SIZE + corrected FETCH_SIZE, collected as MI355X_MICROARCH.md prescribes) OF THE KERNEL INSTANCE THIS ROW TIMED:
  a record counts only if it names the same kernel, the same launch shape (pcx_engine_launch_shape; the tuners pick per
  box) and the same API (single steps / step_n) -- (bytes, the record's source), or (None, None)."""
  path = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
  try:
    for rec in json.load(open(path)).get('records', []):
      if (rec.get('game') == game and rec.get('level') == level and rec.get('batch') == batch and rec.get('kernel') == kernel and
          rec.get('launch_shape') == launch_shape and rec.get('api', 'step') == api and rec.get('source')):
        return rec['bytes_per_launch'], rec['source']
  except Exception:  # pylint: disable=broad-except
    pass
  return None, None


def settle_tuner(eng, row, start, limit=96):
  """Untimed: steps until the engine's launch-shape tuner has settled (include/pcx.h pcx_engine_tuner_done: 8 + 24 launches
  for the persistent workers, 6 + 12 for pcx_generic_step's waves per workgroup), so that no timed repeat contains a
  measuring launch.  Returns the number of extra steps taken."""
  import torch
  extra = 0
  while extra < limit:
    if eng.tuner_done():
      break
    for _ in range(4):
      eng.step(row(start + extra))
      extra += 1
    torch.cuda.synchronize()  # (the tuner settles on a completed event, polled at the next launch)
  return extra


def launch_floor_us(device):
  """What one dependent kernel launch costs on this box: a near-empty kernel launched back to back on one stream."""
  import ctypes
  import torch
  from pycolab_amd import _native as N
  buf = torch.empty(1 << 16, dtype=torch.uint8, device='cuda:%d' % device)
  stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
  best = float('inf')
  for _ in range(3):
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(1000):
      N.lib().pcx_device_fill_probe(buf.data_ptr(), 1024, stream)
    e1.record()
    torch.cuda.synchronize(device)
    best = min(best, e0.elapsed_time(e1))
  return best  # ms per 1000 launches = us per launch


TRAFFIC_SOURCE = ('profiles/hbm_traffic.json (committed rocprofv3 --pmc passes of this kernel at this batch; '
                  'a constant looked up by workload, NOT measured in this run)')


def spawn_ranks(n, argv, oversubscribe):
  """`--gpus N` without a launcher: become `torch.distributed.run` with N ranks."""
  import torch
  have = torch.cuda.device_count()
  if have < 1 or (have < n and not oversubscribe):
    raise SystemExit('bench.py: --gpus %d but this node exposes %d GPU(s)' % (n, have))
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
         '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv
  os.execv(sys.executable, cmd)


def time_steps(eng, row, lo, hi, barrier, after_step=None):
  """Times steps [lo, hi) -- actions row(t) -- once: (wall seconds, avg ms per step on the launch stream)."""
  import torch
  barrier()
  ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  trace = [] if os.environ.get('PCX_BENCH_TRACE') else None  # (diagnosis only: an event after every launch, printed to stderr)
  t0 = time.perf_counter()
  ev0.record()
  for t in range(lo, hi):
    eng.step(row(t))
    if after_step is not None:
      after_step()
    if trace is not None:
      trace.append(torch.cuda.Event(enable_timing=True))
      trace[-1].record()
  ev1.record()
  barrier()
  wall = time.perf_counter() - t0
  if trace:
    marks = [ev0] + trace
    sys.stderr.write('steps [%d, %d) per launch, ms: %s\n' % (lo, hi, ' '.join('%.3f' % marks[i].elapsed_time(marks[i + 1]) for i in range(len(trace)))))
  return wall, ev0.elapsed_time(ev1) / (hi - lo)


def median(xs):
  s = sorted(xs)
  return s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2])


def fill_probe_gbs(nbytes, device):
  """This box's store-only bandwidth, same run: best of 8 launches of pcx_device_fill_probe
  over as many bytes as the step kernel writes per launch."""
  import ctypes
  import torch
  from pycolab_amd import _native as N
  nbytes = max(1 << 24, nbytes // 4 * 4)
  buf = torch.empty(nbytes, dtype=torch.uint8, device='cuda:%d' % device)
  stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
  best = float('inf')
  for i in range(10):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    N.check(N.lib().pcx_device_fill_probe(buf.data_ptr(), nbytes, stream))
    ev1.record()
    torch.cuda.synchronize(device)
    if i >= 2:
      best = min(best, ev0.elapsed_time(ev1))
  del buf
  return nbytes / (best * 1e-3) / 1e9


def measure_config(game, level, batch, steps, warmup, device, repeats=3, raises=None, cardinal_fields=0, cpu_seconds=0.0):
  """One of the other BASELINE configs on this GPU (reported inside the headline line).
  cardinal_fields = n: the game's action is n four-bit fields (one per scrolling group: oracle/walker_scenarios.py), and
  the synthetic tape draws every field from {north, east, south, west, stay} -- as oracle/gen_golden.py does for the
  reference trace of the same game -- instead of uniformly from all 16 values: uniform draws mix diagonal and
  out-of-table motions into conflicting scroll orders, for which the reference raises (protocols/scrolling.py:372-434);
  a third of the environments of VERDICT r4's run sat in that state, never terminating, and were timed anyway."""
  import torch
  from pycolab_amd import _native as N
  from pycolab_amd.compiler import GameTemplate
  from pycolab_amd.engine import Engine
  fixture = FIXTURES[game] % level if '%' in FIXTURES[game] else FIXTURES[game]
  template = GameTemplate.load(os.path.join(ROOT, 'tests', 'golden', 'templates', fixture + '.npz'))
  forced = game.endswith('_generic') and os.environ.get('PCX_FORCE_GENERIC') is None
  if forced:
    os.environ['PCX_FORCE_GENERIC'] = '1'
  try:
    eng = Engine.from_template(template, batch=batch, device=device, auto_reset=True, seed=0x5EED)
    eng.its_showtime()  # (the native engine -- and with it the kernel -- is chosen here)
  finally:
    if forced:
      del os.environ['PCX_FORCE_GENERIC']
  g = torch.Generator(device='cuda')
  g.manual_seed(0x5EED)
  if cardinal_fields:
    values = torch.tensor([0, 2, 4, 6, 8], dtype=torch.int32, device='cuda')  # n, e, s, w, stay (walker_scenarios.MOTION_NAMES)
    tape = torch.zeros((warmup + steps, batch), dtype=torch.int32, device='cuda')
    for f in range(cardinal_fields):
      tape |= values[torch.randint(0, 5, (warmup + steps, batch), device='cuda', generator=g)] << (4 * f)
  else:
    tape = torch.randint(0, template.n_actions, (warmup + steps, batch), dtype=torch.int32, device='cuda', generator=g)
  for t in range(warmup):
    eng.step(tape[t])
  settled = settle_tuner(eng, lambda t: tape[t % warmup], 0)
  sync = lambda: torch.cuda.synchronize(device)
  runs = [time_steps(eng, lambda t: tape[t], warmup, warmup + steps, sync)[1] for _ in range(repeats)]
  kernel_ms = median(runs)
  if raises is None:
    eng.check_errors()
  bps = int(N.lib().pcx_engine_bytes_per_step(eng._native))
  where = 'examples' if game in ('scrolly_maze', 'warehouse', 'marauders', 'hello_world', 'better_scrolly_maze') else 'tests/golden/templates'
  kernel, shape = N.lib().pcx_engine_kernel_name(eng._native).decode(), int(N.lib().pcx_engine_launch_shape(eng._native))
  traffic, source = pmc_traffic(game, level, batch, kernel, shape)
  out = {'workload': '%s/%s, %d envs' % (where, fixture, batch), 'ms_per_step': kernel_ms,
         'ms_per_step_min_max': [min(runs), max(runs)],
         'env_steps_per_s': batch / (kernel_ms * 1e-3),
         'kernel': kernel, 'launch_shape': shape, 'tuner_steps_before_timing': settled, 'algorithmic_bytes_per_env_step': bps,
         'hbm_frac': bps * batch / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
         'traffic': traffic, 'traffic_source': source}
  if cpu_seconds > 0 and reference_dir() is not None:  # the reference itself on this box's host cores, same workload
    out['cpu_baseline'] = cpu_reference_python(game, level, seconds=cpu_seconds)
  if cardinal_fields:
    out['tape'] = '%d action fields, each drawn from {n, e, s, w, stay}' % cardinal_fields
    out['environments_that_raised'] = {'count': int(eng.buffers['error'].tensor.ne(0).sum())}  # (check_errors() above: none)
  if raises is not None:  # a game that raises under uniform random actions (the reference would too): flagged environments keep stepping
    out['environments_that_raised'] = {'count': int(eng.buffers['error'].tensor.ne(0).sum()), 'why': raises}
  eng.close()
  return out


def measure_story(batch, steps, device):
  """SURVEY 8 f-4's cited game, examples/ordeal.py, as ONE batched `storytelling.Story` (every environment in its own chapter;
  three chapter engines stepped by pcx_generic_step, chapter changes and plot words carried on the host): the 16 stories recorded
  from the reference (tests/golden/traces/ordeal_story.npz: data) tiled over the batch, timed per play() with the host side
  included; the last row is compared with the trace."""
  import numpy as np
  import torch
  from pycolab_amd import cropping, storytelling
  from pycolab_amd.compiler import GameTemplate
  from pycolab_amd.engine import Engine
  tr = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'traces', 'ordeal_story.npz')))
  T, E = tr['actions'].shape
  T = min(T, steps)
  keys = ('castle', 'cavern', 'kansas')
  load = lambda k: GameTemplate.load(os.path.join(ROOT, 'tests', 'golden', 'templates', 'ordeal_%s.npz' % k))
  story = storytelling.Story(chapters={k: (lambda k=k: Engine.from_template(load(k), batch=batch, device=device)) for k in keys},
                             croppers=dict(castle=None, cavern=None, kansas=cropping.ScrollingCropper(rows=8, cols=15, to_track='P', scroll_margins=(2, 3))),
                             first_chapter='kansas', auto_reset=True)  # examples/ordeal.py:82-110 make_game()
  tile = np.arange(batch) % E
  actions = torch.from_numpy(np.ascontiguousarray(tr['actions'][:T, tile])).to('cuda:%d' % device)
  story.its_showtime()
  torch.cuda.synchronize(device)
  t0 = time.perf_counter()
  for t in range(T):
    obs, reward, discount = story.play(actions[t])
  torch.cuda.synchronize(device)
  wall = time.perf_counter() - t0
  equal = bool(np.array_equal(obs.board.cpu().numpy(), tr['boards'][T][tile]) and np.array_equal(reward, tr['reward'][T][tile]) and
               np.array_equal(discount, tr['discount'][T][tile]))
  story.close()
  return {'workload': 'examples/ordeal.py as one batched storytelling.Story (tests/golden/templates/ordeal_*), %d envs, %d steps of the '
                      'recorded tapes' % (batch, T), 'ms_per_step': wall / T * 1e3, 'env_steps_per_s': batch * T / wall,
          'kernel': 'pcx_generic_step x 3 chapter engines + host-side chapter changes', 'last_row_equals_reference_trace': equal,
          'note': 'host-bound: one launch per live chapter per play(), chapter changes decided on the host'}


def guarded(fn, *args, **kw):
  """A secondary row of the line must not cost the headline its line: what went wrong stands in its place."""
  try:
    return fn(*args, **kw)
  except Exception as e:  # pylint: disable=broad-except
    return {'workload': '%s%r' % (fn.__name__, args[:3]), 'error': '%s: %s' % (type(e).__name__, str(e)[:300])}


def measure_step_n(game, level, batch, steps, device, repeats=3):
  """BASELINE config 2 through `Engine.step_n(tape)`: the launches take several steps each at this batch size (the
  cooperative instance walks them with the state words in registers; include/pcx.h pcx_engine_step_n) -- every step
  still writes its full observation.  ms per step by HIP events around the whole tape."""
  import torch
  from pycolab_amd import _native as N
  from pycolab_amd import device as pdev
  from pycolab_amd.compiler import GameTemplate
  from pycolab_amd.engine import Engine
  fixture = FIXTURES[game] % level if '%' in FIXTURES[game] else FIXTURES[game]
  template = GameTemplate.load(os.path.join(ROOT, 'tests', 'golden', 'templates', fixture + '.npz'))
  eng = Engine.from_template(template, batch=batch, device=device, auto_reset=True, seed=0x5EED)
  eng.its_showtime()
  g = torch.Generator(device='cuda')
  g.manual_seed(0x5EED)
  tape = torch.randint(0, template.n_actions, (steps, batch), dtype=torch.int32, device='cuda', generator=g)
  stream = pdev.current_stream(device)
  run = lambda: N.check(N.lib().pcx_engine_step_n(eng._native, tape.data_ptr(), steps, 1, stream))
  run()
  kernel, shape = N.lib().pcx_engine_kernel_name(eng._native).decode(), int(N.lib().pcx_engine_launch_shape(eng._native))
  traffic, source = pmc_traffic(game, level, batch, kernel, shape, api='step_n')
  runs = []
  for _ in range(repeats):
    torch.cuda.synchronize(device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    run()
    ev1.record()
    torch.cuda.synchronize(device)
    runs.append(ev0.elapsed_time(ev1) / steps)
  ms = median(runs)
  eng.check_errors()
  bps = int(N.lib().pcx_engine_bytes_per_step(eng._native))
  how = ('several steps per launch' if shape in (12, 13) else
         'single-step launches (the engine walks steps inside a launch only where that pays: up to 655,360 environments, profiles/r06_stepn_crossover.txt)')
  out = {'workload': 'examples/%s, %d envs, Engine.step_n(tape of %d steps): %s, every step writes its '
                     'observation' % (fixture, batch, steps, how),
         'ms_per_step': ms, 'ms_per_step_min_max': [min(runs), max(runs)], 'env_steps_per_s': batch / (ms * 1e-3),
         'kernel': kernel, 'launch_shape': shape, 'algorithmic_bytes_per_env_step': bps,
         'hbm_frac': bps * batch / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': source}
  eng.close()
  return out


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=200)
  ap.add_argument('--warmup', type=int, default=20)
  ap.add_argument('--repeats', type=int, default=5,
                  help='how many times the K timed steps are repeated; value = the median repeat')
  ap.add_argument('--batch', type=int, default=1 << 20,
                  help='environments per GPU (--scaling weak) or in total (--scaling strong)')
  ap.add_argument('--scaling', default=None, choices=['weak', 'strong'],
                  help='default: strong (fixed global batch, BASELINE config 5) for --gpus > 1, weak for one GPU')
  ap.add_argument('--gather', action='store_true', default=None,
                  help='also time steps followed by the all-gather of the packed 10 B/env step results (default for --gpus > 1)')
  ap.add_argument('--no-gather', dest='gather', action='store_false')
  ap.add_argument('--oversubscribe', action='store_true',
                  help='allow more ranks than GPUs (ranks share devices; process group over gloo)')
  ap.add_argument('--actions', default='random', choices=['random', 'hashed'],
                  help='random: a device-generated uniform tape per rank (default).  hashed: pcx_action_hash(0x5EED, '
                       'GLOBAL env, step) %% n_actions, staged from the host -- what an unsharded '
                       'Engine.step_hashed(0x5EED, 0, T) draws, so shards can be checked against one engine')
  ap.add_argument('--dump-scalars', default=None, metavar='NPZ',
                  help='after the last step: all-gather reward/reward_set/discount/done (ScalarGather) and per-rank '
                       'observation checksums, and have rank 0 save them with the number of steps taken')
  ap.add_argument('--level', type=int, default=0)
  ap.add_argument('--game', default='scrolly_maze', choices=sorted(g for g in FIXTURES if not g.endswith('_generic')),
                  help='scrolly_maze is the headline metric; the others are the parity configs of BASELINE.json')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-other-configs', action='store_true',
                  help='skip the short measurements of BASELINE configs 2-4 added to the N=1 headline line')
  args = ap.parse_args()
  if args.scaling is None:
    args.scaling = 'strong' if args.gpus > 1 else 'weak'
  if args.gather is None:
    args.gather = args.gpus > 1
  if args.repeats < 1 or args.steps < 1:
    raise SystemExit('bench.py: --steps and --repeats must be >= 1')

  launched = 'RANK' in os.environ and 'MASTER_ADDR' in os.environ  # by torch.distributed.run
  if args.gpus > 1 and not launched:
    spawn_ranks(args.gpus, sys.argv[1:], args.oversubscribe)  # does not return
  rank = int(os.environ.get('RANK', '0')) if launched else 0
  world = int(os.environ.get('WORLD_SIZE', '1')) if launched else 1
  local = int(os.environ.get('LOCAL_RANK', '0')) if launched else 0
  if args.gpus != world:
    raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))

  import torch
  n_dev = torch.cuda.device_count()
  if n_dev <= local and not (args.oversubscribe and n_dev >= 1):
    raise SystemExit('bench.py: rank %d has no GPU (device_count=%d)' % (rank, n_dev))
  device = local % n_dev
  oversubscribed = world > n_dev
  torch.cuda.set_device(device)
  distributed = world > 1 or launched
  backend = None
  if distributed:
    import torch.distributed as dist
    if oversubscribed:
      backend = 'gloo'
      dist.init_process_group('gloo')
    else:
      backend = 'nccl'  # = RCCL on ROCm
      dist.init_process_group('nccl', device_id=torch.device('cuda', device))
    assert dist.get_world_size() == world

  from pycolab_amd import _native as N
  from pycolab_amd import actions as pactions
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
  eng = Engine.from_template(template, batch=B, device=device, auto_reset=True, seed=0x5EED, env_offset=lo)
  eng.its_showtime()

  # Synthetic action tape, resident in HBM before timing: uniform ordinary
  # actions {0..n_actions-1}, one int32 row per step.
  K, W, R = args.steps, args.warmup, args.repeats
  total = W + K * (R + (1 if args.gather and distributed else 0))
  if args.actions == 'hashed':  # every step has its own row: step t of global env e plays hash(seed, e, t)
    tape = torch.from_numpy(pactions.hashed_tape(0x5EED, lo, B, 0, total, template.n_actions)).to('cuda:%d' % device)
    row = lambda t: tape[t]
  else:  # W + K rows; every repeat replays rows [W, W + K)
    g = torch.Generator(device='cuda')
    g.manual_seed(0x5EED + rank)
    tape = torch.randint(0, template.n_actions, (W + K, B), dtype=torch.int32, device='cuda', generator=g)
    row = lambda t: tape[t if t < W else W + (t - W) % K]

  def barrier():
    if distributed:
      dist.barrier()
    torch.cuda.synchronize()

  def max_over_ranks(x):
    if not distributed:
      return x
    w = torch.tensor([x], dtype=torch.float64, device='cpu' if backend == 'gloo' else 'cuda')
    dist.all_reduce(w, op=dist.ReduceOp.MAX)
    return float(w.item())

  def every_rank(x):
    if not distributed:
      return [x]
    out = [None] * world
    dist.all_gather_object(out, x)
    return out

  # proof that the collective backend spans all N ranks (a SCALE record must be able to show RCCL saw N): every rank
  # contributes its rank from its own device, one all_gather_into_tensor, count the distinct stamps that arrived
  ranks_seen = None
  if distributed:
    dev_t = 'cpu' if backend == 'gloo' else 'cuda:%d' % device
    stamp = torch.tensor([rank], dtype=torch.int32, device=dev_t)
    got_stamps = torch.full((world,), -1, dtype=torch.int32, device=dev_t)
    dist.all_gather_into_tensor(got_stamps, stamp)
    ranks_seen = int(torch.unique(got_stamps[got_stamps >= 0]).numel())

  for t in range(W):
    eng.step(row(t))
  # the launch-shape tuner measures on the engine's own first launches: to completion before anything is timed (every rank
  # takes the same number of extra steps; they replay warm-up rows)
  settled = settle_tuner(eng, lambda t: row(t % W if W else 0), 0) if W else 0
  if distributed:
    most = int(max_over_ranks(float(settled)))
    for t in range(settled, most):
      eng.step(row(t % W))
  walls, kernels = [], []
  for r in range(R):
    wall, kernel_ms = time_steps(eng, row, W + r * K, W + (r + 1) * K, barrier)
    walls.append(max_over_ranks(wall))
    kernels.append(kernel_ms)
  steps_taken = W + R * K
  eng.check_errors()
  wall = median(walls)
  kernel_ms = median(kernels)
  per_rank_kernel_ms = every_rank(kernel_ms)

  gather = None
  if args.gather and distributed:
    sg = pdist.ScalarGather(eng.scalars_packed, global_batch=global_batch)
    sg.gather()  # communicator warm-up
    gwall, _ = time_steps(eng, row, steps_taken, steps_taken + K, barrier, after_step=sg.gather)
    steps_taken += K
    gwall = max_over_ranks(gwall)
    gather = {'collective': 'all_gather_into_tensor over %s, one per step, 10 B/env packed record%s'
                            % ('RCCL' if backend == 'nccl' else 'gloo (staged through pinned host memory)',
                               ', no host sync' if backend == 'nccl' else ''),
              'bytes_per_rank_per_step': 10 * B, 'ms_per_step_with_gather': gwall / K * 1e3,
              'value_with_gather': global_batch * K / gwall}

  if args.dump_scalars:
    # the union of the shards, as a consumer would gather it, + a checksum of every rank's observation
    torch.cuda.synchronize()
    planes = eng.buffers['planes'].tensor.view(B, -1)
    per_env = planes.sum(dim=1, dtype=torch.int64)
    weights = torch.arange(lo + 1, lo + B + 1, dtype=torch.int64, device=planes.device)
    checks = every_rank({'rank': rank, 'lo': lo, 'n': B, 'device': device,
                         'planes_sum': int(per_env.sum().item()), 'planes_weighted': int((per_env * weights).sum().item()),
                         'frame_sum': int(eng.buffers['frame'].tensor.sum(dtype=torch.int64).item())})
    if distributed:
      sg = pdist.ScalarGather(eng.scalars_packed, global_batch=global_batch)
      sg.gather()
      got = [x.cpu().numpy() for x in sg.unpack()]
    else:
      got = [eng.buffers[k].numpy() for k in ('reward', 'reward_set', 'discount', 'done')]
    if rank == 0:
      import numpy as np
      np.savez(args.dump_scalars, reward=got[0], reward_set=got[1], discount=got[2], done=got[3],
               steps_taken=np.array([steps_taken]), checks=np.frombuffer(json.dumps(checks).encode(), np.uint8))

  per_rank_envs = [pdist.shard_range(global_batch, r, world)[1] - pdist.shard_range(global_batch, r, world)[0]
                   if args.scaling == 'strong' else args.batch for r in range(world)]

  # N > 1 with the fixed global batch as the headline: the weak-scaling figure (args.batch environments on EVERY GPU) is a
  # second timed block of the same line, on a second engine
  weak = None
  if distributed and world > 1 and args.scaling == 'strong':
    engw = Engine.from_template(template, batch=args.batch, device=device, auto_reset=True, seed=0x5EED, env_offset=rank * args.batch)
    engw.its_showtime()
    gw = torch.Generator(device='cuda')
    gw.manual_seed(0x5EED + 1000 + rank)
    tapew = torch.randint(0, template.n_actions, (W + K, args.batch), dtype=torch.int32, device='cuda', generator=gw)
    for t in range(W):
      engw.step(tapew[t])
    ww, wk = [], []
    for r in range(min(R, 3)):
      wall_w, kms_w = time_steps(engw, lambda t: tapew[t], W, W + K, barrier)
      ww.append(max_over_ranks(wall_w))
      wk.append(kms_w)
    engw.check_errors()
    bps_w = int(N.lib().pcx_engine_bytes_per_step(engw._native))
    per_rank_w = every_rank(median(wk))
    weak = {'scaling': 'weak', 'batch_per_gpu': args.batch, 'global_batch': args.batch * world, 'steps': K,
            'value': args.batch * world * K / median(ww), 'unit': 'env-steps/s', 'ms_per_step': median(ww) / K * 1e3,
            'per_rank_kernel_ms': per_rank_w,
            'hbm_frac_per_rank': [bps_w * args.batch / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS for ms in per_rank_w]}
    del tapew
    engw.close()

  if rank == 0:
    bytes_per_step = int(N.lib().pcx_engine_bytes_per_step(eng._native))
    achieved = bytes_per_step * B / (kernel_ms * 1e-3) / 1e9
    kernel_name, shape_now = N.lib().pcx_engine_kernel_name(eng._native).decode(), int(N.lib().pcx_engine_launch_shape(eng._native))
    traffic, traffic_src = pmc_traffic(args.game, args.level, B, kernel_name, shape_now)
    L = len(template.chars)
    line = {
        'metric': 'env-steps/sec (whole node), scrolly_maze batch=1M; bit-exact vs CPU' if args.game == 'scrolly_maze'
                  else 'env-steps/sec (whole node), %s' % fixture,
        'value': global_batch * K / wall,
        'unit': 'env-steps/s',
        'n_gpus': world,
        'steps': K,
        'warmup': W,
        'ms_per_step': wall / K * 1e3,
        'higher_is_better': True,
        'scaling': args.scaling,
        'vs_baseline': None,
        'dtype': 'u8',
        'data': 'synthetic',
        'config': {'workload': 'examples/%s level %d, %d envs per GPU, uniform actions 0-%d, '
                               'auto-reset episodes, full observation (board + %d layers) every step'
                               % (fixture, args.level, B, template.n_actions - 1, L),
                   'batch_per_gpu': B, 'global_batch': global_batch, 'parallelism': 'env-shard x%d' % world},
        'repeats': {'k': R, 'statistic': 'median', 'steps_each': K,
                    'ms_per_step_all': [w / K * 1e3 for w in walls],
                    'ms_per_step_min': min(walls) / K * 1e3, 'ms_per_step_max': max(walls) / K * 1e3,
                    'kernel_ms_all': kernels},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                     'traffic_source': (TRAFFIC_SOURCE + ': ' + traffic_src) if traffic is not None else None,
                     'kernel': kernel_name, 'launch_shape': shape_now, 'tuner_steps_before_timing': settled,
                     'kernel_ms': kernel_ms, 'algorithmic_bytes_per_env_step': bytes_per_step},
    }
    if world > 1:
      # N > 1: `achieved` / `frac` above are RANK 0's GPU (one GPU's HBM against one GPU's peak); every rank's own figure:
      fr = [bytes_per_step * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS for n, ms in zip(per_rank_envs, per_rank_kernel_ms)]
      line['roofline']['scope'] = 'per GPU (rank 0); frac_per_rank lists every rank'
      line['roofline']['frac_per_rank'] = fr
      line['roofline']['frac_min'] = min(fr)
      line['roofline']['frac_median'] = median(fr)
    if distributed:
      line['dist'] = {'backend': backend, 'world_size': dist.get_world_size(), 'oversubscribed': oversubscribed,
                      'devices_on_node': n_dev, 'rank_device': every_rank_devices(world, n_dev),
                      'per_rank_kernel_ms': per_rank_kernel_ms, 'per_rank_envs': per_rank_envs,
                      # (nccl IS RCCL on ROCm; gloo only under --oversubscribe, where RCCL refuses two ranks per device)
                      'collective_ranks_seen': ranks_seen,
                      'rccl_version': list(torch.cuda.nccl.version()) if backend == 'nccl' else None}
    if weak is not None:
      line['weak_scaling'] = weak
    if gather is not None:
      line['gather'] = gather
    del tape
    eng.close()
    if not oversubscribed:
      # what this box's HBM takes from a kernel that only stores, same run: the
      # step kernel's bytes are 96 % writes, so this is its practical ceiling
      plane_bytes = B * (1 + L) * template.rows * template.cols
      fill = fill_probe_gbs(min(plane_bytes, 8 << 30), device)
      line['roofline']['achievable'] = {'GBps': fill, 'what': 'pcx_device_fill_probe: store-only kernel over the '
                                        'step kernel\'s output size, best of 8 launches, same run'}
      line['roofline']['frac_of_achievable'] = achieved / fill
    if world == 1 and args.game == 'scrolly_maze' and not args.no_other_configs:
      # BASELINE configs 2-4 on the same GPU, same run (their own kernels and rooflines), then the other two
      # hand-written kernels: SURVEY 8 f-1 (better_scrolly_maze, 45x89 board) and config 1's game on the GPU
      # (+ config 5's per-GPU shard sizes: 131,072 environments = 1,048,576 over eight GPUs, and 262,144 = over four)
      line['other_configs'] = [guarded(measure_config, 'scrolly_maze', 0, 131072, 200, 20, device),
                               guarded(measure_config, 'scrolly_maze', 0, 262144, 200, 20, device),
                               guarded(measure_config, 'scrolly_custom_H', 0, 131072, 200, 20, device),
                               guarded(measure_config, 'scrolly_maze', 1, 131072, 200, 20, device),
                               guarded(measure_config, 'scrolly_maze', 0, 4096, 200, 20, device),
                               guarded(measure_step_n, 'scrolly_maze', 0, 4096, 1000, device),
                               # the headline batch through Engine.step_n: launches of up to 64 steps in which every persistent
                               # worker keeps its units from step to step (launch shape 13); every step still writes its observation
                               guarded(measure_step_n, 'scrolly_maze', 0, 131072, 256, device),
                               guarded(measure_step_n, 'scrolly_maze', 0, 1048576, 128, device),
                               guarded(measure_config, 'marauders', 0, 32768, 200, 20, device, cpu_seconds=0 if args.no_cpu_baseline else 4.0),
                               guarded(measure_config, 'marauders', 0, 262144, 50, 10, device),
                               guarded(measure_config, 'warehouse', 0, 262144, 100, 10, device, cpu_seconds=0 if args.no_cpu_baseline else 4.0),
                               guarded(measure_config, 'better_scrolly_maze', 0, 65536, 50, 10, device),
                               guarded(measure_config, 'hello_world', 0, 1048576, 50, 10, device),
                               # pcx_generic_step (built for the template at run time: launch_shape 31) at VERDICT r3's fixtures
                               guarded(measure_config, 'marauders_custom_A', 0, 32768, 200, 30, device),
                               guarded(measure_config, 'walkers_scroll_groups', 0, 262144, 100, 30, device, cardinal_fields=2),
                               guarded(measure_config, 'warehouse_generic', 0, 262144, 100, 30, device),
                               guarded(measure_config, 'ordeal_kansas', 0, 262144, 100, 30, device),
                               guarded(measure_story, 65536, 160, device)]
      # config 2's 11 us per play() against what a launch costs on this box at all
      line['launch_floor_us'] = {'value': guarded(launch_floor_us, device), 'what': 'a near-empty kernel (pcx_device_fill_probe over 1 KiB), 1,000 '
                                 'launches back to back on one stream, best of three: the floor under config 2\'s ms_per_step'}
    if not args.no_cpu_baseline:  # (rank 0's host cores, N > 1 included)
      # north_star: "the reference CPU Engine timed on the same box's host cores (core count stated) in the same run" --
      # the imported reference where it is available (kind "reference"), with the C restatement of it ("port") next to
      # it; the port alone where it is not
      port = guarded(cpu_baseline, template_path)
      ref = guarded(cpu_reference_python, args.game, args.level) if args.game in ('scrolly_maze', 'warehouse', 'marauders', 'hello_world', 'better_scrolly_maze') else None
      if ref is not None and 'error' in ref and 'error' not in port:  # (the reference leg failed: the port alone, as where oracle/_ref is absent)
        sys.stderr.write('bench.py: the reference leg of the CPU baseline failed: %s\n' % ref['error'])
        ref = None
      if ref is not None:
        line['cpu_baseline'] = ref
        line['cpu_baseline_port'] = port
      else:
        line['cpu_baseline'] = port
    print(json.dumps(line))
  if distributed:
    dist.barrier()
    dist.destroy_process_group()


def every_rank_devices(world, n_dev):
  return [r % n_dev for r in range(world)]


if __name__ == '__main__':
  main()
