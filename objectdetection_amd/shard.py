"""Per-image sharding across ranks: one process per GPU, no data-path collective.

Images are the independent unit of the whole path (the reference maps
patch_batch per example, rpn_train.py:450, and runs NMS / crop per batch
element, module/model.py:51-53), so rank r simply owns a contiguous slice of
the global batch.  torch.distributed is used only to learn rank / world size and,
optionally, to gather small per-image results for reporting.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch


def rank_world() -> Tuple[int, int]:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_images: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, stop) of the images rank `rank` owns; the first n % world ranks get one extra."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(n_images, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard(tensors: Sequence[torch.Tensor], rank: int, world: int) -> List[torch.Tensor]:
    """Slice every [B,...] tensor to this rank's images."""
    n = tensors[0].shape[0]
    s, e = shard_range(n, rank, world)
    return [t[s:e] for t in tensors]


def gather_small(t: torch.Tensor, n_images: int) -> torch.Tensor:
    """Reassemble a small per-image tensor ([b_local, ...]) on every rank, in image
    order (reporting only; ragged shards are padded to the largest)."""
    import torch.distributed as dist
    rank, world = rank_world()
    if world == 1:
        return t
    sizes = [shard_range(n_images, r, world) for r in range(world)]
    mx = max(e - s for s, e in sizes)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[: e - s] for b, (s, e) in zip(bufs, sizes)], dim=0)
