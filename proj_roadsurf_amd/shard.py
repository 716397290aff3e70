"""Tile sharding for multi-GPU inference: tiles are independent units -- the
reference's hot loop is a flat per-dataset list ``for d in DatasetCatalog.get(dataset): predictor(im)``
([EXT od] make_detections.py, R:README.md:78) -- so rank r of R processes (one per GPU) takes a
contiguous block of the list and runs its own engine.  There is NO collective on the data path;
``torch.distributed`` is only used (optionally) to gather the per-tile results on rank 0 for the
single output file per dataset (host objects, works over gloo or RCCL alike).
"""
from __future__ import annotations

from typing import Any, Callable, List, Optional, Sequence, Tuple


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank ``rank``; sizes differ by at most one, earlier ranks get the
    extra item (keeps output files locality-ordered)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} of {world}")
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def run_sharded(items: Sequence[Any], predict_batch: Callable[[Sequence[Any]], List[Any]], batch: int,
                rank: int = 0, world: int = 1, gather: bool = True) -> Optional[List[Any]]:
    """Run ``predict_batch`` over this rank's block in batches of ``batch``; with ``gather`` the
    per-item results of all ranks are returned on rank 0 in the original item order (None elsewhere)."""
    lo, hi = shard_range(len(items), rank, world)
    mine: List[Any] = []
    for i in range(lo, hi, batch):
        mine.extend(predict_batch(items[i:min(i + batch, hi)]))
    if len(mine) != hi - lo:
        raise RuntimeError(f"predict_batch returned {len(mine)} results for {hi - lo} items")
    if world == 1 or not gather:
        return mine
    import torch.distributed as dist

    parts: Optional[List[Any]] = [None] * world if rank == 0 else None
    dist.gather_object(mine, parts, dst=0)
    if rank != 0:
        return None
    out: List[Any] = []
    for p in parts:          # rank order == item order for contiguous blocks
        out.extend(p)
    return out
