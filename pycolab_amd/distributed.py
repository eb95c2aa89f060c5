"""Sharding the batch over the GPUs of one node (one process per GPU).

Environments never interact (reference: one `Engine` per environment,
engine.py:102-104), so the batch is cut into contiguous ranges and every rank
steps its own range with no data-path collective.  The only optional exchange
is a gather of the per-environment scalars `play()` returns (reward,
reward_set, discount, done: 10 bytes per environment).  The engine keeps those
four arrays in ONE device allocation (`Engine.scalars_packed`), so the gather
is a single `all_gather_into_tensor` of that buffer -- RCCL over xGMI on GPUs
(backend "nccl"), gloo in the CPU tests -- with no packing pass, no host
synchronisation and no per-call allocation.
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


def pack_scalars(reward, reward_set, discount, done):
  """The engine's packed layout [reward i32 | discount f32 | reward_set u8 |
  done u8] from four separate 1-D tensors (one concatenation)."""
  import torch
  return torch.cat([reward.contiguous().view(torch.uint8), discount.contiguous().view(torch.uint8),
                    reward_set.contiguous().view(torch.uint8), done.contiguous().view(torch.uint8)])


def unpack_scalars(packed, n):
  """Typed zero-copy views (reward, reward_set, discount, done) of one packed
  block of `n` environments."""
  import torch
  return (packed[0:4 * n].view(torch.int32), packed[8 * n:9 * n],
          packed[4 * n:8 * n].view(torch.float32), packed[9 * n:10 * n])


class ScalarGather(object):
  """All-gather of every rank's packed step results, one collective per call.

  `packed`: this rank's uint8 tensor [10 * n_local] (`Engine.scalars_packed`);
  it is read in place at every `gather()`.  `global_batch` fixes every rank's
  shard length through `shard_range` (no size exchange per step); when omitted
  all shards must have this rank's length -- checked ONCE here with a size
  exchange, so that ragged shards raise instead of hanging the collective.
  The receive buffer is allocated once.

  Device tensors travel over the group's own backend ("nccl" = RCCL over
  xGMI).  A group whose backend cannot take device tensors (gloo: the
  `bench.py --oversubscribe` mode, where several ranks share one GPU and RCCL
  refuses duplicate devices) stages the 10 B/env block through pinned host
  memory -- the same collective, shapes and unpacking.
  """

  def __init__(self, packed, global_batch=None, group=None):
    import torch
    import torch.distributed as dist
    self._dist, self._group = dist, group
    self.world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_local = packed.numel() // 10
    if global_batch is None:
      sizes = [None] * self.world
      dist.all_gather_object(sizes, int(n_local), group=group)
      if len(set(sizes)) != 1:
        raise ValueError('shards differ in length ({}): pass global_batch so that every rank derives the same '
                         'layout from shard_range'.format(sizes))
      self.sizes = [n_local] * self.world
    else:
      self.sizes = [hi - lo for lo, hi in (shard_range(global_batch, r, self.world) for r in range(self.world))]
      if self.sizes[rank] != n_local:
        raise ValueError('rank {} holds {} environments, shard_range says {}'.format(rank, n_local, self.sizes[rank]))
    self._local = packed
    self._slot = (10 * max(self.sizes) + 15) // 16 * 16  # bytes per rank in the receive buffer (typed views stay aligned)
    self._send = packed
    if packed.numel() != self._slot:   # a shorter (or oddly sized) shard sends a padded copy of its block
      self._send = torch.zeros(self._slot, dtype=torch.uint8, device=packed.device)
    self.out = torch.empty(self.world * self._slot, dtype=torch.uint8, device=packed.device)
    self.staged = bool(packed.is_cuda and dist.get_backend(group) == 'gloo')
    if self.staged:
      self._host_send = torch.zeros(self._slot, dtype=torch.uint8).pin_memory()
      self._host_out = torch.empty(self.world * self._slot, dtype=torch.uint8).pin_memory()

  def gather(self):
    """Issues the collective on the current stream; returns the raw receive
    buffer [world, slot] (asynchronous on GPUs over RCCL: no host sync here)."""
    if self._send is not self._local:
      self._send[:self._local.numel()].copy_(self._local)
    if self.staged:
      self._host_send.copy_(self._send)  # device -> pinned host, synchronous
      self._dist.all_gather_into_tensor(self._host_out, self._host_send, group=self._group)
      self.out.copy_(self._host_out, non_blocking=True)
    else:
      self._dist.all_gather_into_tensor(self.out, self._send, group=self._group)
    return self.out.view(self.world, self._slot)

  def unpack(self):
    """(reward, reward_set, discount, done) of the whole batch in global
    environment order, from the last `gather()`."""
    import torch
    blocks = self.out.view(self.world, self._slot)
    parts = [unpack_scalars(blocks[r], n) for r, n in enumerate(self.sizes)]
    return tuple(torch.cat([p[i] for p in parts]) for i in range(4))


def gather_scalars(reward, reward_set, discount, done, group=None, global_batch=None):
  """One-shot form for four separate tensors: packs them (one concatenation),
  gathers with ONE collective and returns the four arrays of the whole batch
  in global environment order.  Shards of different length need
  `global_batch` (sizes then follow `shard_range`)."""
  g = ScalarGather(pack_scalars(reward, reward_set, discount, done), global_batch=global_batch, group=group)
  g.gather()
  return g.unpack()
