"""Utterance sharding over the GPUs of one node.

The path has no cross-utterance operation, so the batch axis is split contiguously over ranks, weights are
replicated, noise is keyed on the GLOBAL utterance index, and the only exchange is one all_gather of the
per-clip scores ([B_local, K] fp32, ~20 KB per rank) at the end.  `torch.distributed` (backend "nccl" = RCCL on
ROCm, "gloo" in the CPU tests) carries it; there is no collective inside the sampling loop.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n_utts: int, rank: int, world: int):
    """Contiguous, balanced split: the first (n % world) ranks get one extra clip. -> (start, stop)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(n_utts, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_gather_scores(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Gather [B_local, K] scores of every rank into [n_total, K] in global utterance order (uneven shards are
    padded to the largest shard for the collective and trimmed afterwards)."""
    if not (dist.is_available() and dist.is_initialized()):
        if local.shape[0] != n_total:
            raise ValueError("single process but local batch != total")
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    if local.shape[0] != sizes[rank][1] - sizes[rank][0]:
        raise ValueError(f"rank {rank}: local batch {local.shape[0]} != shard size {sizes[rank][1] - sizes[rank][0]}")
    mx = max(e - s for s, e in sizes)
    buf = local.new_zeros((mx,) + tuple(local.shape[1:]))
    buf[: local.shape[0]] = local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    return torch.cat([p[: e - s] for p, (s, e) in zip(parts, sizes)], dim=0)
