"""TEST INFRASTRUCTURE: times the imported reference itself (google-deepmind/pycolab
under /root/reference) on the host cores -- `bench.py`'s `cpu_reference_python`
leg, which exists only where the reference does (the build container; the GPU
box has no /root/reference).

Every worker process loops ONE reference `Engine` (examples/<game>.make_game)
over its own environment indices with the synthetic workload of SURVEY 8(d):
action = pcx_action_hash(0x5EED, env, t) % n_actions, and an environment whose
episode ended is rebuilt (`make_game(); its_showtime()`, one env-step) -- the
policy the GPU path and the C oracle run.  Stepping only is timed."""
import os
import sys
import time
import warnings

GAMES = {  # bench.py --game -> (module, make_game takes a level, ordinary actions)
    'scrolly_maze': ('scrolly_maze', True, 5),
    'warehouse': ('warehouse_manager', True, 5),
    'marauders': ('extraterrestrial_marauders', False, 4),
    'hello_world': ('hello_world', False, 4),
    'better_scrolly_maze': ('better_scrolly_maze', True, 5),
}


def _worker(args):
  reference, game, level, seconds, index = args
  sys.dont_write_bytecode = True
  sys.path.insert(0, reference)
  warnings.filterwarnings('ignore', category=DeprecationWarning)
  import importlib
  from oracle import binding
  module, levelled, n_actions = GAMES[game]
  mod = importlib.import_module('pycolab.examples.' + module)
  make = (lambda: mod.make_game(level)) if levelled else mod.make_game
  env = index  # global environment index of this worker's (single) environment
  g = make()
  g.its_showtime()
  steps, t = 0, 0
  t0 = time.perf_counter()
  while True:
    if g.game_over:
      g = make()
      g.its_showtime()
    else:
      g.play(int(binding.action_hash(0x5EED, env, t) % n_actions))
    t += 1
    steps += 1
    if steps % 256 == 0 and time.perf_counter() - t0 >= seconds:
      break
  return steps, time.perf_counter() - t0


def measure(reference, game, level, seconds=10.0, max_procs=64):
  import multiprocessing as mp
  cores = max(1, min(max_procs, len(os.sched_getaffinity(0))))
  with mp.get_context('fork').Pool(cores) as pool:
    res = pool.map(_worker, [(reference, game, level, seconds, i) for i in range(cores)])
  steps = sum(r[0] for r in res)
  wall = max(r[1] for r in res)
  return {'value': steps / wall, 'unit': 'env-steps/s', 'cores': cores, 'kind': 'reference',
          'per_core': steps / wall / cores,
          'sample': 'the imported Python reference (pycolab.examples.%s.make_game -> engine.py its_showtime / play, from %s), %d procs x one Engine each, '
                    '%.1f s of stepping, hashed actions + rebuild-on-game-over' % (GAMES[game][0], reference, cores, wall)}


if __name__ == '__main__':
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  import json
  print(json.dumps(measure(os.environ.get('PCX_REFERENCE', '/root/reference'), sys.argv[1] if len(sys.argv) > 1 else 'scrolly_maze',
                           int(sys.argv[2]) if len(sys.argv) > 2 else 0, seconds=float(sys.argv[3]) if len(sys.argv) > 3 else 5.0)))
