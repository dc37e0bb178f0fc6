"""Pair sharding across the GPUs of one node.

Registration pairs are independent (GroupNorm is per sample, BatchNorm is in
eval mode), so the path shards embarrassingly: static partition of the pair
list, weights replicated, NO data-path collective.  The only exchange is one
all_gather of the (R,t) results per shard (<= 240 B per pair) — RCCL over xGMI
on GPUs (backend "nccl"), gloo in the CPU tests.  The reference has no
distributed code at all; this is the multi-GPU analogue of its per-pair loop
(reference test.py:386-450).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch


def shard_range(num_items: int, rank: int, world: int) -> range:
    """Contiguous block partition; the first (num_items % world) ranks get one extra item."""
    base, rem = divmod(int(num_items), int(world))
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def shard_sizes(num_items: int, world: int) -> List[int]:
    return [len(shard_range(num_items, r, world)) for r in range(world)]


def gather_results(local: torch.Tensor, dist=None, sizes: Optional[Sequence[int]] = None) -> torch.Tensor:
    """all_gather per-pair results along dim 0.  ``sizes`` = number of valid rows
    per rank when shards are unequal (rows are padded to the maximum for the
    collective and trimmed afterwards).  With ``dist`` None or no process group this
    is the identity; with an initialised group the collective RUNS, also at world
    size 1 (one rank gathering from itself: the same RCCL call path as N ranks)."""
    if dist is None or not dist.is_initialized():
        return local if sizes is None else local[: sizes[0]]
    world = dist.get_world_size()
    on_gpu = local.is_cuda
    if on_gpu and dist.get_backend() == "gloo":     # CPU rehearsal backend: stage through host memory
        return gather_results(local.cpu(), dist, sizes).to(local.device)
    if sizes is None:
        out = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(out, local.contiguous())
        return torch.cat(out, 0)
    pad_to = max(sizes)
    buf = local.new_zeros((pad_to,) + tuple(local.shape[1:]))
    buf[: local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return torch.cat([o[:s] for o, s in zip(out, sizes)], 0)
