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
    # gloo (the CPU tests, and the one-device test hook of bench.py) moves host memory only: device
    # tensors are bounced there.  Under RCCL the buffers never leave HBM.
    bounce = local.is_cuda and dist.get_backend(group) != "nccl"
    wire_dev = torch.device("cpu") if bounce else local.device
    wire_local = local.cpu() if bounce else local
    ops = []
    receivers = range(world) if dst is None else [dst]
    for r in receivers:
        if r == rank:
            for s in range(world):
                if s != rank:
                    parts[s] = torch.empty(int(sizes[s]), dtype=torch.uint8, device=wire_dev)
                    if sizes[s]:
                        ops.append(dist.P2POp(dist.irecv, parts[s], s, group))
        elif sizes[rank]:
            ops.append(dist.P2POp(dist.isend, wire_local, r, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if bounce:
        parts = [p if p is None or p is local else p.to(local.device) for p in parts]
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


# ---- device-buffer variants (no host bounce between encode and gather) ---------------------------
# The numpy functions above exist for callers that hold host arrays (and for the CPU tests, which
# inject the oracle).  On GPUs the payload is produced in HBM and RCCL moves HBM buffers, so these
# variants take and return torch tensors on the rank's device and never call .cpu():
# source rows are SCATTERED from the rank that holds the surface (grouped send / receive of exact
# block-row ranges -- the "trivial block-range scatter" of north_star, Converter.cpp:521-527 being
# the serial surface loop that is distributed), every rank encodes with cfhip_encode_device on the
# current torch stream, and the payload ranges are gathered the same way.

def scatter(parts, sizes: Sequence[int], rank: int, world: int, src: int = 0, device=None,
            group=None):
    """Exact-size scatter of uint8 tensors: on rank `src`, parts[r] (sizes[r] bytes, a device
    tensor or a view of one) goes to rank r; every rank returns its own part.  Grouped
    point-to-point transfers (RCCL: one ncclGroup of sends), buffers stay on the device."""
    import torch
    import torch.distributed as dist

    bounce = world > 1 and dist.get_backend(group) != "nccl"     # gloo test hook: host memory on the wire
    if rank == src:
        assert parts is not None and all(int(parts[r].numel()) == int(sizes[r]) for r in range(world))
        mine = parts[rank]
        ops = [dist.P2POp(dist.isend, parts[r].cpu() if bounce else parts[r].contiguous(), r, group)
               for r in range(world) if r != src and sizes[r]]
    else:
        mine = torch.empty(int(sizes[rank]), dtype=torch.uint8, device="cpu" if bounce else device)
        ops = [dist.P2POp(dist.irecv, mine, src, group)] if sizes[rank] else []
    if world > 1 and ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if bounce and rank != src and device is not None:
        mine = mine.to(device)
    return mine


def row_ranges(height: int, bh: int, world: int):
    """Per rank: (first scanline, end scanline, first block row, end block row) of its shard."""
    rows = (height + bh - 1) // bh
    out = []
    for r in range(world):
        a, b = api.shard_rows(rows, r, world)
        out.append((a * bh, min(b * bh, height), a, b))
    return out


def encode_rows_sharded_device(ctx, image, width: int, height: int, pixel_type, params, rank: int,
                               world: int, src: Optional[int] = 0, dst: Optional[int] = 0,
                               group=None, timings: Optional[dict] = None):
    """ONE surface split by block rows across the ranks, device buffers end to end.

    image: on rank `src` a contiguous uint8 torch tensor holding the whole surface (height rows of
    width * texel bytes) on this rank's GPU; ignored elsewhere.  src=None: every rank already holds
    the surface (or at least its own rows) in `image` -- no scatter.  Returns the payload as a
    uint8 device tensor: the whole surface's on rank `dst` (every rank for dst=None), this rank's
    rows elsewhere.  Byte-identical to the one-rank encode (the last shard keeps the true bottom
    edge).  `timings` (optional dict) receives scatter / encode / gather seconds of this rank."""
    import time as _time
    import torch

    bw, bh, bs = block_dims(params)
    texel = {int(api.PixelType.RGBA8): 4, int(api.PixelType.RGBA32F): 16, int(api.PixelType.RGBA16F): 8}[int(pixel_type)]
    row_bytes = width * texel
    bx = (width + bw - 1) // bw
    ranges = row_ranges(height, bh, world)
    dev = torch.device("cuda", torch.cuda.current_device())
    sync = torch.cuda.synchronize
    t0 = _time.perf_counter()
    y0, y1, a, b = ranges[rank]
    if src is None or world == 1:
        flat = image.reshape(-1)
        local_src = flat[y0 * row_bytes:y1 * row_bytes]
    else:
        sizes = [(r1 - r0) * row_bytes for (r0, r1, _, _) in ranges]
        parts = None
        if rank == src:
            flat = image.reshape(-1)
            parts = [flat[r0 * row_bytes:r1 * row_bytes] for (r0, r1, _, _) in ranges]
        local_src = scatter(parts, sizes, rank, world, src, dev, group)
    if timings is not None:
        sync()
        timings["scatter_s"] = _time.perf_counter() - t0
        t0 = _time.perf_counter()
    out_sizes = [max(0, rb - ra) * bx * bs for (_, _, ra, rb) in ranges]
    local = torch.empty(out_sizes[rank], dtype=torch.uint8, device=dev)
    if b > a:
        ctx.encode_device([{"pixels": local_src.data_ptr(), "pixel_type": int(pixel_type), "width": width,
                            "height": y1 - y0, "row_pitch_bytes": row_bytes, "out": local.data_ptr(),
                            "out_capacity": out_sizes[rank]}], params, _producer_stream(torch))
    if timings is not None:
        sync()
        timings["encode_s"] = _time.perf_counter() - t0
        t0 = _time.perf_counter()
    if world == 1:
        return local
    torch.cuda.current_stream().synchronize()      # RCCL runs on its own stream
    parts = exchange(local, out_sizes, rank, world, dst, group)
    if timings is not None:
        sync()
        timings["gather_s"] = _time.perf_counter() - t0
    if dst is None or dst == rank:
        return torch.cat([p for p in parts])
    return local


def _producer_stream(torch) -> int:
    """The stream handle to give the library so that it reads what torch (and RCCL, which hands its
    results to torch's current stream) produced.  A real stream orders the kernels by itself.  The
    legacy default stream has handle 0, which the C ABI reads as "the context's own stream" -- a
    non-blocking one that does not wait for the default stream -- so finish the producers first."""
    h = torch.cuda.current_stream().cuda_stream
    if not h:
        torch.cuda.synchronize()
    return h


def encode_surfaces_sharded_device(ctx, surfaces: Sequence[dict], params, rank: int, world: int,
                                   dst: Optional[int] = 0, group=None):
    """A batch of device-resident surfaces across the ranks (LPT on block counts; every rank holds
    or can produce its own units).  surfaces: dicts with width / height / pixel_type and, for the
    units this rank owns, pixels (a uint8 device tensor) -- other units' pixels may be None.
    Returns, per surface, its payload as a uint8 device tensor on rank `dst` (all ranks for
    dst=None) and on the owner; None elsewhere.  One batched encode per rank, one exact-size
    exchange, no host bounce."""
    import torch

    bw, bh, _ = block_dims(params)
    counts = [block_count(s["width"], s["height"], bw, bh) for s in surfaces]
    nbytes = [api.payload_size(params.format, params.type, s["width"], s["height"]) for s in surfaces]
    plan = assign_surfaces(counts, world)
    mine = plan[rank]
    dev = torch.device("cuda", torch.cuda.current_device())
    sizes = [sum(nbytes[i] for i in plan[r]) for r in range(world)]
    local = torch.empty(sizes[rank], dtype=torch.uint8, device=dev)
    texel = {int(api.PixelType.RGBA8): 4, int(api.PixelType.RGBA32F): 16, int(api.PixelType.RGBA16F): 8}
    desc, off = [], 0
    for i in mine:
        s = surfaces[i]
        pitch = s.get("row_pitch_bytes", s["width"] * texel[int(s["pixel_type"])])
        desc.append({"pixels": s["pixels"].data_ptr(), "pixel_type": int(s["pixel_type"]), "width": s["width"],
                     "height": s["height"], "row_pitch_bytes": pitch, "out": local.data_ptr() + off,
                     "out_capacity": nbytes[i]})
        off += nbytes[i]
    if desc:
        order = sorted(range(len(desc)), key=lambda k: desc[k]["pixel_type"])     # one launch per source type
        ctx.encode_device([desc[k] for k in order], params, _producer_stream(torch))
    torch.cuda.current_stream().synchronize()
    parts = exchange(local, sizes, rank, world, dst, group) if world > 1 else [local]
    result: list = [None] * len(surfaces)
    for r, part in enumerate(parts):
        if part is None:
            continue
        off = 0
        for i in plan[r]:
            result[i] = part[off:off + nbytes[i]]
            off += nbytes[i]
    return result
