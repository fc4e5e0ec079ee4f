"""Multi-GPU sharding of the block-encode path.

Every block depends only on its own texels (S3tcConverter.cpp:242-255, AstcConverter.cpp:208-225)
and surfaces are independent (Converter.cpp:521-589), so the path shards with NO data-path
collective:

* a batch of surfaces (texture arrays / mip chains, BASELINE config 5) is split by unit
  (a surface, or a whole texture with its chain) with a deterministic longest-processing-time
  assignment on block counts;
* one large surface is split by contiguous block rows (``cfhip_shard_rows``) of the FORMAT's
  block height (4 for BC / ETC, 4..12 for the ASTC footprints).

One process per GPU.  ``torch.distributed`` (RCCL on GPUs, gloo in the CPU tests) is used only
for the optional gather of the payload -- the "trivial block-range gather" of north_star.  The
plan is a pure function of the sizes, so every rank knows every rank's byte count: the gather is
a grouped exact-size send / receive of DEVICE buffers (no size exchange, no padding to the
largest share, no host bounce on RCCL).  The encoder is injected
(``encode_fn(images, params) -> list of uint8 arrays``): the product passes ``Context.encode``;
the CPU tests pass the oracle.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np

from . import api


def block_dims(params) -> tuple:
    """(block width, block height, bytes per block) of the params' (format, type)."""
    return api.query(params.format, params.type)


def block_count(width: int, height: int, bw: int = 4, bh: int = 4) -> int:
    return ((width + bw - 1) // bw) * ((height + bh - 1) // bh)


def assign_surfaces(block_counts: Sequence[int], world: int) -> List[List[int]]:
    """Deterministic LPT: units sorted by (blocks desc, index asc) go to the least loaded
    rank (ties -> lowest rank).  Returns, per rank, its unit indices in ascending order."""
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


def exchange(local, sizes: Sequence[int], rank: int, world: int, dst: Optional[int] = None,
             group=None) -> list:
    """Exact-size gather of one uint8 torch tensor per rank (sizes[r] bytes on rank r, known to
    all ranks from the plan).  dst=None: every rank receives every part; dst=k: rank k only.
    Buffers stay on the tensors' device (GPU under RCCL); grouped point-to-point transfers.
    Returns the list of parts (None where this rank does not receive)."""
    import torch
    import torch.distributed as dist

    assert local.dtype == torch.uint8 and local.numel() == sizes[rank]
    parts: list = [None] * world
    parts[rank] = local
    if world == 1:
        return parts
    ops = []
    receivers = range(world) if dst is None else [dst]
    for r in receivers:
        if r == rank:
            for s in range(world):
                if s != rank:
                    parts[s] = torch.empty(int(sizes[s]), dtype=torch.uint8, device=local.device)
                    if sizes[s]:
                        ops.append(dist.P2POp(dist.irecv, parts[s], s, group))
        elif sizes[rank]:
            ops.append(dist.P2POp(dist.isend, local, r, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return parts


def _dist_device():
    import torch
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def encode_surfaces_sharded(images: Sequence[np.ndarray], params, encode_fn: Callable,
                            rank: int, world: int, gather: bool = True,
                            group=None) -> List[Optional[np.ndarray]]:
    """Encode a batch of surfaces across ``world`` ranks.  With gather=True every rank gets
    every payload (byte-identical to a 1-rank run); otherwise only its own (others None)."""
    import torch

    bw, bh, _ = block_dims(params)
    counts = [block_count(im.shape[1], im.shape[0], bw, bh) for im in images]
    plan = assign_surfaces(counts, world)
    nbytes = [api.payload_size(params.format, params.type, im.shape[1], im.shape[0]) for im in images]
    mine = plan[rank]
    outs = encode_fn([images[i] for i in mine], params) if mine else []
    result: List[Optional[np.ndarray]] = [None] * len(images)
    for i, o in zip(mine, outs):
        result[i] = np.asarray(o, np.uint8).reshape(-1)
    if gather and world > 1:
        local = np.concatenate([result[i] for i in mine]) if mine else np.zeros(0, np.uint8)
        sizes = [sum(nbytes[i] for i in plan[r]) for r in range(world)]
        parts = exchange(torch.from_numpy(local).to(_dist_device()), sizes, rank, world, None, group)
        for r, part in enumerate(parts):
            if r == rank:
                continue
            host = part.cpu().numpy()
            off = 0
            for i in plan[r]:
                result[i] = host[off:off + nbytes[i]].copy()
                off += nbytes[i]
    return result


def encode_rows_sharded(image: np.ndarray, params, encode_fn: Callable, rank: int, world: int,
                        gather: bool = True, group=None) -> Optional[np.ndarray]:
    """Encode ONE surface split by block rows (cfhip_shard_rows).  Rank r owns the block rows
    [a, b): contiguous source scanlines and a contiguous slice of the payload."""
    import torch

    h, w = image.shape[0], image.shape[1]
    bw, bh, bs = block_dims(params)
    rows = (h + bh - 1) // bh
    a, b = api.shard_rows(rows, rank, world)
    local = np.zeros(0, np.uint8)
    if b > a:
        # the last shard keeps the true bottom edge so edge replication is unchanged
        local = np.asarray(encode_fn([image[a * bh:min(b * bh, h)]], params)[0], np.uint8).reshape(-1)
    if not gather or world == 1:
        return local
    row_bytes = ((w + bw - 1) // bw) * bs
    sizes = []
    for r in range(world):
        ra, rb = api.shard_rows(rows, r, world)
        sizes.append(max(0, rb - ra) * row_bytes)
    parts = exchange(torch.from_numpy(local).to(_dist_device()), sizes, rank, world, None, group)
    return np.concatenate([p.cpu().numpy() for p in parts])
