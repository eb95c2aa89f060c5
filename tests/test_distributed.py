"""world_size-2 gloo test of the multi-GPU path's host logic: env sharding by
rank + the optional scalar gather.  The per-rank engines here are the CPU
oracle (test infrastructure) standing in for the HIP engine, so the test runs
without a GPU; the union of the two shards must equal one unsharded run."""
import os
import socket
import sys

import numpy as np
import pytest

from pycolab_amd import distributed as pdist
from tests import helpers


def test_shard_range_partitions():
  for batch in (1, 7, 64, 1000, 1 << 20):
    for world in (1, 2, 3, 8):
      spans = [pdist.shard_range(batch, r, world) for r in range(world)]
      assert spans[0][0] == 0 and spans[-1][1] == batch
      assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
      sizes = [hi - lo for lo, hi in spans]
      assert max(sizes) - min(sizes) <= 1
  with pytest.raises(ValueError):
    pdist.shard_range(10, 2, 2)


def _worker(rank, world, port, batch, steps, out_dir):
  sys.path.insert(0, helpers.ROOT)
  import torch
  import torch.distributed as dist
  from oracle import binding
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  lo, hi = pdist.shard_range(batch, rank, world)
  eng = binding.OracleEngine(helpers.load_template('scrolly_maze_L0'), hi - lo)
  eng.reset()
  eng.step_hashed(0xABCD, 0, steps, env_offset=lo)   # global env index drives the actions
  got = pdist.gather_scalars(torch.from_numpy(np.array(eng.reward)), torch.from_numpy(np.array(eng.reward_set)),
                             torch.from_numpy(np.array(eng.discount)), torch.from_numpy(np.array(eng.done)),
                             global_batch=batch)
  if rank == 0:
    np.savez(os.path.join(out_dir, 'gathered.npz'), reward=got[0].numpy(), reward_set=got[1].numpy(),
             discount=got[2].numpy(), done=got[3].numpy())
  dist.barrier()
  dist.destroy_process_group()


def test_two_rank_shards_equal_one_unsharded_run(tmp_path):
  import torch.multiprocessing as mp
  from oracle import binding
  batch, steps, world = 333, 40, 2   # odd batch: shards of different length
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  mp.spawn(_worker, args=(world, port, batch, steps, str(tmp_path)), nprocs=world, join=True)
  got = np.load(str(tmp_path / 'gathered.npz'))
  whole = binding.OracleEngine(helpers.load_template('scrolly_maze_L0'), batch)
  whole.reset()
  whole.step_hashed(0xABCD, 0, steps)
  for name in ('reward', 'reward_set', 'discount', 'done'):
    np.testing.assert_array_equal(got[name], np.array(getattr(whole, name)), err_msg=name)
  assert got['done'].sum() > 0


@pytest.mark.gpu
def test_rccl_gather_of_device_scalars_single_rank():
  """The gather runs on the engine's own device tensors over the "nccl"
  backend (= RCCL on ROCm).  One GPU here, so the group has one rank; two
  engines stand in for two shards (env_offset carries the global index) and
  their concatenation must equal one unsharded engine."""
  import torch
  import torch.distributed as dist
  from tests.hip_adapter import HipAdapter
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  torch.cuda.set_device(0)
  dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
  try:
    t = helpers.load_template('scrolly_maze_L0')
    batch, steps = 1000, 48
    whole = HipAdapter(t, batch); whole.reset(); whole.step_hashed(0xABCD, 0, steps)
    parts = []
    for rank in range(2):
      lo, hi = pdist.shard_range(batch, rank, 2)
      eng = HipAdapter(t, hi - lo, env_offset=lo); eng.reset(); eng.step_hashed(0xABCD, 0, steps, env_offset=lo)
      tensors = [eng.eng.buffers[n].tensor for n in ('reward', 'reward_set', 'discount', 'done')]
      assert all(x.is_cuda for x in tensors)
      parts.append([g.cpu().numpy() for g in pdist.gather_scalars(*tensors)])  # world of 1: returns this shard
    for i, name in enumerate(('reward', 'reward_set', 'discount', 'done')):
      np.testing.assert_array_equal(np.concatenate([parts[0][i], parts[1][i]]), whole.read(name), err_msg=name)
  finally:
    dist.destroy_process_group()


def _run_bench(*flags):
  import subprocess
  env = dict(os.environ)
  for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
    env.pop(k, None)
  return subprocess.run([sys.executable, os.path.join(helpers.ROOT, 'bench.py')] + list(flags), env=env,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)


def test_bench_refuses_more_gpus_than_the_node_has():
  """`bench.py --gpus N` spawns its own N ranks; with fewer devices it must
  fail loudly instead of benchmarking one GPU and printing n_gpus: 1."""
  import torch
  n = torch.cuda.device_count() + 1 if torch.cuda.device_count() else 2
  r = _run_bench('--gpus', str(max(n, 2)), '--steps', '2', '--warmup', '1', '--no-cpu-baseline')
  assert r.returncode != 0
  assert 'GPU(s)' in r.stderr and '"n_gpus"' not in r.stdout


def _ragged_worker(rank, world, port, out_dir):
  sys.path.insert(0, helpers.ROOT)
  import torch
  import torch.distributed as dist
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  n = 5 + rank  # shards of different length and no global_batch: must raise on every rank, not hang
  try:
    pdist.ScalarGather(torch.zeros(10 * n, dtype=torch.uint8))
    verdict = 'no error'
  except ValueError as e:
    verdict = 'ValueError: %s' % e
  open(os.path.join(out_dir, 'rank%d.txt' % rank), 'w').write(verdict)
  dist.barrier()
  dist.destroy_process_group()


def test_ragged_shards_without_global_batch_raise(tmp_path):
  import torch.multiprocessing as mp
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  mp.spawn(_ragged_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
  for rank in range(2):
    verdict = open(str(tmp_path / ('rank%d.txt' % rank))).read()
    assert verdict.startswith('ValueError') and 'global_batch' in verdict, verdict


def test_hashed_tape_is_the_device_generator():
  """pycolab_amd.actions.hashed_tape (what a sharded bench stages) == pcx_action_hash of the C ABI."""
  from pycolab_amd import _native as N
  from pycolab_amd import actions
  tape = actions.hashed_tape(0x5EED, 1000, 7, 3, 5, 5)
  want = [[N.lib().pcx_action_hash(0x5EED, 1000 + e, 3 + t) % 5 for e in range(7)] for t in range(5)]
  np.testing.assert_array_equal(tape, np.array(want, np.int32))


@pytest.mark.gpu
@pytest.mark.parametrize('scaling,batch,ranks', [('strong', 1001, 2), ('weak', 512, 2), ('strong', 10007, 8), ('weak', 300, 8)])
def test_bench_n_ranks_equal_one_unsharded_engine(tmp_path, scaling, batch, ranks):
  """The N>1 path of bench.py, executed: two and EIGHT ranks (ragged shards: 10,007 environments over eight) spawned by bench.py itself through
  torch.distributed.run -- over RCCL where the node has two GPUs; on a one-GPU box with
  `--oversubscribe` (both ranks share the GPU, so the group is gloo; launcher, shard_range /
  env_offset, weak and strong accounting, ScalarGather with padded sends and the rank-0 JSON
  are the real path's).  The union of the two shards -- gathered scalars and per-rank
  observation checksums -- must equal ONE unsharded engine."""
  import json
  import torch
  from tests.hip_adapter import HipAdapter
  dump = str(tmp_path / 'gathered.npz')
  real = torch.cuda.device_count() >= ranks
  r = _run_bench('--gpus', str(ranks), *([] if real else ['--oversubscribe']), '--steps', '5', '--warmup', '2', '--repeats', '2',
                 '--batch', str(batch), '--scaling', scaling, '--gather', '--actions', 'hashed',
                 '--dump-scalars', dump, '--no-cpu-baseline')
  assert r.returncode == 0, r.stderr[-3000:]
  line = json.loads(r.stdout.strip().splitlines()[-1])
  global_batch = batch if scaling == 'strong' else ranks * batch
  sizes = [hi - lo for lo, hi in (pdist.shard_range(global_batch, k, ranks) for k in range(ranks))]
  assert line['n_gpus'] == ranks and line['scaling'] == scaling
  assert line['config']['global_batch'] == global_batch and line['config']['batch_per_gpu'] == sizes[0]
  assert line['dist']['world_size'] == ranks and line['dist']['collective_ranks_seen'] == ranks and line['dist']['backend'] == ('nccl' if real else 'gloo')
  assert line['dist']['oversubscribed'] == (not real)
  assert line['dist']['per_rank_envs'] == sizes and len(line['dist']['per_rank_kernel_ms']) == ranks
  assert all(ms > 0 for ms in line['dist']['per_rank_kernel_ms'])
  assert line['repeats']['k'] == 2 and len(line['repeats']['ms_per_step_all']) == 2
  assert line['gather']['bytes_per_rank_per_step'] == 10 * sizes[0]
  assert abs(line['value'] - global_batch * 5 / (line['ms_per_step'] * 5e-3)) < 1e-6 * line['value']
  got = np.load(dump)
  steps = int(got['steps_taken'][0])
  assert steps == 2 + 2 * 5 + 5
  whole = HipAdapter(helpers.load_template('scrolly_maze_L0'), global_batch)
  whole.reset()
  whole.step_hashed(0x5EED, 0, steps)
  for name in ('reward', 'reward_set', 'discount', 'done'):
    np.testing.assert_array_equal(got[name], whole.read(name), err_msg=name)
  planes = whole.read('planes').reshape(global_batch, -1).astype(np.int64).sum(axis=1)
  frames = whole.read('frame').astype(np.int64)
  checks = json.loads(bytes(got['checks']).decode())
  assert [c['rank'] for c in checks] == list(range(ranks)) and [c['n'] for c in checks] == sizes
  for c in checks:
    lo, hi = c['lo'], c['lo'] + c['n']
    assert c['planes_sum'] == int(planes[lo:hi].sum())
    assert c['planes_weighted'] == int((planes[lo:hi] * np.arange(lo + 1, hi + 1)).sum())
    assert c['frame_sum'] == int(frames[lo:hi].sum())
  assert (frames < steps).any()  # episodes ended and restarted inside the run


@pytest.mark.gpu
def test_packed_scalars_gather_matches_buffers():
  """Engine.scalars_packed is the storage behind reward/discount/reward_set/
  done: one all-gather of it (world of one here) unpacks to the same arrays."""
  import torch
  import torch.distributed as dist
  from tests.hip_adapter import HipAdapter
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  torch.cuda.set_device(0)
  dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
  try:
    hip = HipAdapter(helpers.load_template('marauders'), 1003)  # odd size: the send block is padded
    hip.reset(); hip.step_hashed(0xABCD, 0, 64)
    sg = pdist.ScalarGather(hip.eng.scalars_packed, global_batch=1003)
    sg.gather()
    got = sg.unpack()
    for i, name in enumerate(('reward', 'reward_set', 'discount', 'done')):
      np.testing.assert_array_equal(got[i].cpu().numpy(), hip.read(name), err_msg=name)
    assert hip.read('reward_set').any()
  finally:
    dist.destroy_process_group()


@pytest.mark.gpu
def test_c_abi_gather_scalars_over_rccl():
  """include/pcx.h pcx_gather_*: the gather a torch-less FFI host uses -- one process, one RCCL
  communicator per engine's device (one device on this box), bound with dlopen.  Both send
  paths: the engine's own four output arrays (packed by a kernel) and the one packed
  allocation pycolab_amd.Engine binds (sent in place)."""
  import ctypes
  from pycolab_amd import _native as N
  from pycolab_amd.engine import Engine
  lib = N.lib()
  t = helpers.load_template('scrolly_maze_L0')
  steps = 40

  def gathered(native, batch):
    g = ctypes.c_void_p()
    arr = (ctypes.c_void_p * 1)(native)
    N.check(lib.pcx_gather_create(arr, 1, ctypes.byref(g)))
    try:
      N.check(lib.pcx_gather_scalars(g, None))
      recv, slot = ctypes.c_void_p(), N.c_i64()
      N.check(lib.pcx_gather_buffers(g, 0, ctypes.byref(recv), ctypes.byref(slot)))
      assert slot.value == (10 * batch + 15) // 16 * 16
      N.check(lib.pcx_stream_synchronize(None))
      host = np.empty((slot.value,), np.uint8)
      N.check(lib.pcx_memcpy_d2h(host.ctypes.data, recv, host.nbytes))
      return (host[:4 * batch].view(np.int32), host[8 * batch:9 * batch], host[4 * batch:8 * batch].view(np.float32),
              host[9 * batch:10 * batch])
    finally:
      lib.pcx_gather_destroy(g)

  # (a) a bare C-ABI engine: no bound buffers, its own separate output arrays
  batch = 1003
  ct, keep = t.to_ctypes()
  h = ctypes.c_void_p()
  N.check(lib.pcx_engine_create(ctypes.byref(ct), batch, 0, ctypes.byref(h)))
  try:
    N.check(lib.pcx_engine_reset(h, None, None))
    N.check(lib.pcx_engine_step_hashed(h, 0xABCD, 0, 0, steps, 1, None))
    got = gathered(h, batch)
    bufs = N.Buffers()
    N.check(lib.pcx_engine_buffers(h, ctypes.byref(bufs)))
    for i, (name, dt) in enumerate((('reward', np.int32), ('reward_set', np.uint8), ('discount', np.float32), ('done', np.uint8))):
      want = np.empty((batch,), dt)
      N.check(lib.pcx_memcpy_d2h(want.ctypes.data, getattr(bufs, name), want.nbytes))
      np.testing.assert_array_equal(got[i], want, err_msg=name)
    assert got[3].any() or got[1].any()
  finally:
    lib.pcx_engine_destroy(h)
  del keep
  # (b) the Python facade's engine: outputs bound as ONE packed allocation, 10 * batch a multiple of 16
  batch = 1024
  eng = Engine.from_template(t, batch=batch, auto_reset=True)
  eng.its_showtime()
  eng.step_hashed(0xABCD, 0, steps)
  import torch
  torch.cuda.synchronize()
  got = gathered(eng._native, batch)
  for i, name in enumerate(('reward', 'reward_set', 'discount', 'done')):
    np.testing.assert_array_equal(got[i], eng.buffers[name].numpy(), err_msg=name)
  eng.close()


@pytest.mark.gpu
def test_c_abi_gather_two_engines_two_devices():
  """pcx_gather_create over n = 2 engines on two devices of one process (ncclCommInitAll over both; one grouped
  ncclAllGather): every device receives both records, each laid out with its own batch.  Needs two GPUs: the
  one-GPU test box skips it (RCCL refuses two ranks on one device), the 8-GPU node runs it."""
  import ctypes
  import torch
  from pycolab_amd import _native as N
  if torch.cuda.device_count() < 2:
    pytest.skip('pcx_gather_create with n = 2 needs two GPUs (this box has %d); the n = 1 path is '
                'test_c_abi_gather_scalars_over_rccl' % torch.cuda.device_count())
  lib = N.lib()
  t = helpers.load_template('scrolly_maze_L0')
  ct, keep = t.to_ctypes()
  batches, steps = (1003, 517), 40
  engines = []
  for dev, batch in enumerate(batches):
    h = ctypes.c_void_p()
    N.check(lib.pcx_engine_create(ctypes.byref(ct), batch, dev, ctypes.byref(h)))
    N.check(lib.pcx_engine_reset(h, None, None))
    N.check(lib.pcx_engine_step_hashed(h, 0xABCD, 1000 * dev, 0, steps, 1, None))
    engines.append(h)
  g = ctypes.c_void_p()
  try:
    arr = (ctypes.c_void_p * 2)(*engines)
    N.check(lib.pcx_gather_create(arr, 2, ctypes.byref(g)))
    N.check(lib.pcx_gather_scalars(g, None))
    want = []
    for h, batch in zip(engines, batches):
      bufs = N.Buffers()
      N.check(lib.pcx_engine_buffers(h, ctypes.byref(bufs)))
      rec = {}
      for name, dt in (('reward', np.int32), ('discount', np.float32), ('reward_set', np.uint8), ('done', np.uint8)):
        rec[name] = np.empty((batch,), dt)
        N.check(lib.pcx_memcpy_d2h(rec[name].ctypes.data, getattr(bufs, name), rec[name].nbytes))
      want.append(rec)
    for dev in range(2):
      torch.cuda.synchronize(dev)
      recv, slot = ctypes.c_void_p(), N.c_i64()
      N.check(lib.pcx_gather_buffers(g, dev, ctypes.byref(recv), ctypes.byref(slot)))
      assert slot.value == (10 * max(batches) + 15) // 16 * 16
      host = np.empty((2 * slot.value,), np.uint8)
      torch.cuda.set_device(dev)
      N.check(lib.pcx_memcpy_d2h(host.ctypes.data, recv, host.nbytes))
      for r, batch in enumerate(batches):
        rec = host[r * slot.value:(r + 1) * slot.value]
        np.testing.assert_array_equal(rec[:4 * batch].view(np.int32), want[r]['reward'], err_msg='device %d record %d reward' % (dev, r))
        np.testing.assert_array_equal(rec[4 * batch:8 * batch].view(np.float32), want[r]['discount'])
        np.testing.assert_array_equal(rec[8 * batch:9 * batch], want[r]['reward_set'])
        np.testing.assert_array_equal(rec[9 * batch:10 * batch], want[r]['done'])
  finally:
    if g:
      lib.pcx_gather_destroy(g)
    for h in engines:
      lib.pcx_engine_destroy(h)
    torch.cuda.set_device(0)
  del keep
