"""Sharding the batch over the GPUs of one node (one process per GPU).

Environments never interact (reference: one `Engine` per environment,
engine.py:102-104), so the batch is cut into contiguous ranges and every rank
steps its own range with no data-path collective.  The only optional exchange
is a gather of the per-environment scalars `play()` returns (reward,
reward_set, discount, done: 10 bytes per environment), done with
`torch.distributed` -- RCCL over xGMI on GPUs (backend "nccl"), gloo in the
CPU tests.
"""

import os


def shard_range(global_batch, rank, world_size):
  """[lo, hi) of the environments owned by `rank`; sizes differ by at most 1."""
  if not 0 <= rank < world_size:
    raise ValueError('rank {} outside world of {}'.format(rank, world_size))
  base, extra = divmod(int(global_batch), int(world_size))
  lo = rank * base + min(rank, extra)
  return lo, lo + base + (1 if rank < extra else 0)


def env_world():
  """(rank, world_size, local_rank) from the torchrun environment."""
  return (int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')),
          int(os.environ.get('LOCAL_RANK', '0')))


def gather_scalars(reward, reward_set, discount, done, group=None):
  """All-gather the per-environment step results of every rank.

  Arguments are 1-D torch tensors of this rank's shard (any device the
  process group supports).  Shards may differ in length.  Returns the four
  tensors concatenated in rank order -- element i is global environment i.
  """
  import torch
  import torch.distributed as dist
  world = dist.get_world_size(group)
  n = torch.tensor([reward.numel()], dtype=torch.int64, device=reward.device)
  sizes = [torch.zeros_like(n) for _ in range(world)]
  dist.all_gather(sizes, n, group=group)
  sizes = [int(s.item()) for s in sizes]
  longest = max(sizes)
  out = []
  for t in (reward, reward_set, discount, done):
    padded = torch.zeros(longest, dtype=t.dtype, device=t.device)
    padded[:t.numel()] = t
    parts = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    out.append(torch.cat([p[:s] for p, s in zip(parts, sizes)]))
  return tuple(out)
