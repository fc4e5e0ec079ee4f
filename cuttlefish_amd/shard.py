"""Multi-GPU sharding of the block-encode path.

Every 4x4 block depends only on its own texels (S3tcConverter.cpp:242-255) and surfaces are
independent (Converter.cpp:521-589), so the path shards with NO data-path collective:

* a batch of surfaces (texture arrays / mip chains, BASELINE config 5) is split by surface
  with a deterministic longest-processing-time assignment on block counts;
* one large surface is split by contiguous block rows (``cfhip_shard_rows``).

One process per GPU.  ``torch.distributed`` (RCCL on GPUs, gloo in the CPU tests) is used only
for the optional gather of the payload -- the "trivial block-range gather" of north_star.
The encoder is injected (``encode_fn(images, params) -> list of uint8 arrays``): the product
passes ``Context.encode``; the CPU tests pass the oracle.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np

from . import api


def block_count(width: int, height: int) -> int:
    return ((width + 3) // 4) * ((height + 3) // 4)


def assign_surfaces(block_counts: Sequence[int], world: int) -> List[List[int]]:
    """Deterministic LPT: surfaces sorted by (blocks desc, index asc) go to the least loaded
    rank (ties -> lowest rank).  Returns, per rank, its surface indices in ascending order."""
    if world <= 0:
        raise ValueError("world must be positive")
    order = sorted(range(len(block_counts)), key=lambda i: (-int(block_counts[i]), i))
    load = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += int(block_counts[i])
    for lst in out:
        lst.sort()
    return out


def _all_gather_bytes(local: np.ndarray, group=None) -> List[np.ndarray]:
    """All-gather variable-length uint8 arrays (pad to the max length)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else \
        torch.device("cpu")
    n = torch.tensor([local.size], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    buf = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if local.size:
        buf[:local.size] = torch.from_numpy(np.ascontiguousarray(local)).to(dev)
    parts = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    return [p[:s].cpu().numpy() for p, s in zip(parts, sizes)]


def encode_surfaces_sharded(images: Sequence[np.ndarray], params, encode_fn: Callable,
                            rank: int, world: int, gather: bool = True,
                            group=None) -> List[Optional[np.ndarray]]:
    """Encode a batch of surfaces across ``world`` ranks.  With gather=True every rank gets
    every payload (byte-identical to a 1-rank run); otherwise only its own (others None)."""
    counts = [block_count(im.shape[1], im.shape[0]) for im in images]
    plan = assign_surfaces(counts, world)
    mine = plan[rank]
    outs = encode_fn([images[i] for i in mine], params) if mine else []
    result: List[Optional[np.ndarray]] = [None] * len(images)
    for i, o in zip(mine, outs):
        result[i] = o
    if gather and world > 1:
        local = np.concatenate([np.asarray(o, np.uint8).reshape(-1) for o in outs]) \
            if outs else np.zeros(0, np.uint8)
        parts = _all_gather_bytes(local, group)
        for r, part in enumerate(parts):
            off = 0
            for i in plan[r]:
                nbytes = api.payload_size(params.format, params.type, images[i].shape[1],
                                          images[i].shape[0])
                result[i] = part[off:off + nbytes].copy()
                off += nbytes
    return result


def encode_rows_sharded(image: np.ndarray, params, encode_fn: Callable, rank: int, world: int,
                        gather: bool = True, group=None) -> Optional[np.ndarray]:
    """Encode ONE surface split by block rows (cfhip_shard_rows).  Rank r owns the block rows
    [a, b): contiguous source scanlines and a contiguous slice of the payload."""
    h = image.shape[0]
    rows = (h + 3) // 4
    a, b = api.shard_rows(rows, rank, world)
    local = np.zeros(0, np.uint8)
    if b > a:
        # the last shard keeps the true bottom edge so edge replication is unchanged
        local = np.asarray(encode_fn([image[a * 4:min(b * 4, h)]], params)[0], np.uint8)
    if not gather or world == 1:
        return local
    return np.concatenate(_all_gather_bytes(local, group))
