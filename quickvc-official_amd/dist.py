"""Multi-GPU support: utterances shard embarrassingly, one process per GPU.

The reference is single-device and serial over utterances (convert.py:58-86); nothing here
mirrors reference code.  The only collective on the path is ONE broadcast of the packed weight
blob from rank 0 at start-up (RCCL over xGMI with the "nccl" backend; "gloo" in CPU tests).
There is no per-step collective.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process = 1 GPU)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_indices(n_items: int, rank: int, world: int, lengths: Sequence[int] = None) -> List[int]:
    """Static shard of an utterance list: rank r takes items r, r+W, r+2W ... of the list sorted by
    decreasing length (balances ragged corpora; with equal lengths it is plain round-robin)."""
    order = list(range(n_items))
    if lengths is not None:
        order.sort(key=lambda i: (-int(lengths[i]), i))
    return order[rank::world]


def broadcast_blob(blob: torch.Tensor, src: int = 0) -> torch.Tensor:
    """Broadcast the packed weights (uint8 tensor, same size on every rank) from ``src``. In place."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(blob, src=src)
    return blob


def max_over_ranks(value: float, device) -> float:
    """MAX-reduce a scalar (the timed region's wall time) over ranks."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
