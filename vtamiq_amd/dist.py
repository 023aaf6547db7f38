"""Batch-sharded multi-GPU inference: one process per GPU, pairs split contiguously by rank, ONE all-gather of the
predicted scores per step (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).

The reference has no distributed code; the gathered vector reproduces what its validation loop
builds with torch.cat over batches (train.py:403-405): scores in global pair order on every rank.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of the pairs owned by `rank`; the first (global_batch % world) ranks get one extra."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_scores(q_local: torch.Tensor, global_batch: int, group=None, force_collective: bool = False) -> torch.Tensor:
    """All ranks receive q[global_batch] in global pair order.  Uneven shards are padded to the largest shard so the
    collective stays a single fixed-size all-gather.  A single-rank world returns its scores as they are, unless
    `force_collective` (an initialised process group of size 1): then the same all_gather_into_tensor runs on one rank --
    library load, communicator init and stream ordering of the RCCL path exercised on a 1-GPU box (bench.py --force-collective)."""
    initialised = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if initialised else 1
    if world == 1 and not (force_collective and initialised):
        return q_local
    per = -(-global_batch // world)
    send = q_local
    if q_local.numel() != per:
        send = torch.zeros(per, dtype=q_local.dtype, device=q_local.device)
        send[: q_local.numel()] = q_local
    out = torch.empty(world * per, dtype=q_local.dtype, device=q_local.device)
    dist.all_gather_into_tensor(out, send, group=group)
    if global_batch == world * per:
        return out
    parts = []
    for r in range(world):
        lo, hi = shard_range(global_batch, r, world)
        parts.append(out[r * per: r * per + (hi - lo)])
    return torch.cat(parts)


def sharded_forward(model, patches, pos, scales, global_batch: int, group=None) -> torch.Tensor:
    """Run the model on this rank's shard (inputs are the LOCAL shard) and return the gathered global scores."""
    q_local, _ = model(patches, pos, scales)
    return gather_scores(q_local, global_batch, group)
