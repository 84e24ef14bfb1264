"""Frame sharding across ranks: inference frames are independent, so a batch is
split evenly over the GPUs of a node with no data-path collective; ``gather_frames`` exists for
callers that want the outputs back on one rank (simple gather over the process group).

Reference analogue: Lightning's ``DistributedSampler`` split (config/default.py:166).
"""
from __future__ import annotations

import torch


def shard_range(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [lo, hi) slice of ``n_items`` for ``rank``; sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world: {rank}/{world}")
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def scatter_frames(frames: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    lo, hi = shard_range(frames.shape[0], rank, world)
    return frames[lo:hi]


def gather_frames(local: torch.Tensor, n_total: int, group=None, dst: int = 0):
    """Gather per-rank frame shards (possibly of unequal length) onto ``dst``.
    Returns the concatenated tensor on ``dst`` and ``None`` elsewhere."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    maxn = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxn,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    if rank == dst:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.gather(pad, bufs, dst=dst, group=group)
        return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)
    dist.gather(pad, None, dst=dst, group=group)
    return None


def allreduce_gradients(flat: torch.Tensor, group=None) -> float:
    """Training path: ONE sum all-reduce of the flat gradient buffer over the process group (RCCL over xGMI on
    GPUs); returns the factor 1/world that turns the sum into DDP's mean -- the caller fuses it into the clip/Adam kernels
    instead of spending another pass over the buffer.  No-op (factor 1.0) without an initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    world = dist.get_world_size(group)
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world
