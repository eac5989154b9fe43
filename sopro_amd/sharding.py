"""Multi-GPU: utterances are independent (reference: src/sopro/model.py:218-401 has no cross-utterance
term), so a node is N single-GPU engines fed disjoint utterance shards.  There is no data-path
collective: `torch.distributed` is used only to agree on the split, to time (barrier + MAX) and to gather
small host-side results; nothing here touches RCCL/xGMI bandwidth.
"""
from __future__ import annotations

import os
from typing import Any, Callable, List, Optional, Sequence, Tuple


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world) from the torchrun environment (1-process defaults)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin deal: utterance i -> rank i % world (keeps ranks within one utterance of each other
    and preserves arrival order inside a rank)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, int(n_items), world))


def unshard(per_rank: Sequence[Sequence[Any]], n_items: int) -> List[Any]:
    """Inverse of shard_indices over gathered per-rank result lists."""
    world = len(per_rank)
    out: List[Any] = [None] * int(n_items)
    for r, items in enumerate(per_rank):
        idx = shard_indices(n_items, r, world)
        if len(items) != len(idx):
            raise ValueError(f"rank {r} returned {len(items)} results for {len(idx)} utterances")
        for i, v in zip(idx, items):
            out[i] = v
    return out


def run_sharded(items: Sequence[Any], fn: Callable[[List[Any]], List[Any]], *, group: Optional[Any] = None) -> List[Any]:
    """Every rank calls this with the same `items`; `fn` processes the local shard (e.g.
    ``lambda xs: tts.synthesize_batch(...)``) and every rank gets the full, ordered result list back
    (host-side object gather; waveforms are small: 7.7 KB per frame)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return list(fn(list(items)))
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    local = fn([items[i] for i in shard_indices(len(items), rank, world)])
    gathered: List[Any] = [None] * world
    dist.all_gather_object(gathered, list(local), group=group)
    return unshard(gathered, len(items))
