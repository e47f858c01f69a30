"""Batch-sharded sampling over the GPUs of one node.

Samples never interact inside the denoiser or the posterior step, so the only communication is
one all-gather of the finished videos (RCCL over xGMI through ``torch.distributed``'s "nccl"
backend; "gloo" on CPU in tests).  The reference has no distributed code; the single-process API
is unchanged and this module is opt-in.
"""
from __future__ import annotations

from typing import Callable, Tuple

import torch
import torch.distributed as dist


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, stop) of the samples rank `rank` owns; earlier ranks take the remainder."""
    if global_batch < 0 or world <= 0 or not (0 <= rank < world):
        raise ValueError("bad shard arguments")
    base, rem = divmod(global_batch, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_videos(local: torch.Tensor, global_batch: int, group=None) -> torch.Tensor:
    """All-gather the per-rank videos (dim 0 sharded by shard_range) into the full batch on every rank."""
    world = dist.get_world_size(group)
    sizes = [shard_range(global_batch, r, world) for r in range(world)]
    max_n = max(b - a for a, b in sizes)
    pad = torch.zeros((max_n,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[: b - a] for o, (a, b) in zip(out, sizes)], dim=0)


def sample_sharded(sample_local: Callable[[int, int], torch.Tensor], global_batch: int, group=None) -> torch.Tensor:
    """Run `sample_local(start, stop)` (this rank's slice of the batch) and all-gather the results."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    a, b = shard_range(global_batch, rank, world)
    return gather_videos(sample_local(a, b), global_batch, group)
