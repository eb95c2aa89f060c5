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


@pytest.mark.gpu
def test_bench_two_ranks_over_rccl():
  """Two ranks spawned by bench.py itself, weak and strong scaling, with the
  packed scalar all-gather timed (needs a node with at least two GPUs)."""
  import json
  import torch
  if torch.cuda.device_count() < 2:
    pytest.skip('needs two GPUs')
  for scaling, per_gpu in (('weak', 8192), ('strong', 4096)):
    r = _run_bench('--gpus', '2', '--steps', '6', '--warmup', '2', '--batch', '8192', '--scaling', scaling,
                   '--gather', '--no-cpu-baseline')
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['scaling'] == scaling
    assert line['config']['batch_per_gpu'] == per_gpu and line['config']['global_batch'] == 2 * per_gpu
    assert line['gather']['bytes_per_rank_per_step'] == 10 * per_gpu


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
