"""Batch sharding across the GPUs of one node (one process per GPU).

The transforms of a batch are independent (the reference runs them as a
sequential loop of `set_points` + `execute` calls, cc/kernels/nufft_kernels.cc:
491-540), so the batch axis shards with no data-path collective: rank r owns a
contiguous block of items and runs its own plan on its own stream. The only
optional communication is gathering the results (`all_gather`, RCCL over xGMI
when the backend is "nccl"); points shared by the whole batch are broadcast
once by the caller if they do not already live on every rank.
"""
import torch


def shard_bounds(num_items, world_size, rank):
  """Contiguous, balanced split: the first `num_items % world_size` ranks get one extra."""
  base, extra = divmod(int(num_items), int(world_size))
  start = rank * base + min(rank, extra)
  return start, start + base + (1 if rank < extra else 0)


def nufft_sharded(source, points, transform_fn, group=None, gather=True):
  """Applies `transform_fn(source_item_block, points_block)` to this rank's block
  of the leading (batch) axis and optionally all-gathers the results.

  source: [B, ...]; points: [B, M, rank] (per-item) or [M, rank] (shared).
  transform_fn: e.g. lambda s, p: tfft.nufft(s, p, grid_shape=..., transform_type='type_1').
  Returns the full [B, ...] result when gather=True, else this rank's block.
  """
  import torch.distributed as dist
  world = dist.get_world_size(group) if dist.is_initialized() else 1
  rank = dist.get_rank(group) if dist.is_initialized() else 0
  b = source.shape[0]
  lo, hi = shard_bounds(b, world, rank)
  pts = points[lo:hi] if points.dim() == 3 else points
  local = transform_fn(source[lo:hi], pts)
  if not gather or world == 1:
    return local
  return gather_blocks(local, b, group)


def gather_blocks(local, num_items, group=None):
  """all_gather of every rank's contiguous block (as split by shard_bounds) into the full
  [num_items, ...] result on every rank. Ragged blocks are padded to the largest one."""
  import torch.distributed as dist
  world = dist.get_world_size(group)
  rank = dist.get_rank(group)
  lo, hi = shard_bounds(num_items, world, rank)
  maxb = shard_bounds(num_items, world, 0)[1]
  if hi - lo == maxb:
    pad = local.contiguous()
  else:
    pad = torch.zeros((maxb,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:hi - lo] = local
  if pad.is_complex():
    buf = torch.view_as_real(pad).contiguous()
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    outs = [torch.view_as_complex(o) for o in outs]
  else:
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
  parts = []
  for r, o in enumerate(outs):
    s, e = shard_bounds(num_items, world, r)
    parts.append(o[:e - s])
  return torch.cat(parts, dim=0)
