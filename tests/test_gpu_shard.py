"""The sharded encode path with the PRODUCT encoder (Context.encode) on the GPU: world 1 in
process, and world 2 (two ranks, RCCL when two devices are visible, otherwise both ranks share
the device over gloo) -- the N-way result is byte-identical to the 1-way result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cuttlefish_amd import Context, Format, PixelType, Type, make_params, shard, synth

pytestmark = pytest.mark.gpu


def _chain():
    imgs = [synth.photo(64 >> i, 64 >> i, seed=30 + i) for i in range(7)]
    return imgs + [synth.photo(52, 20, seed=40), synth.photo(9, 70, seed=41)]


def test_world1_sharded_equals_direct(gpu_ctx):
    p = make_params(Format.BC7, Type.UNorm, 2)
    imgs = _chain()
    direct = gpu_ctx.encode(imgs, p)
    got = shard.encode_surfaces_sharded(imgs, p, gpu_ctx.encode, 0, 1)
    assert all(np.array_equal(a, b) for a, b in zip(direct, got))
    big = synth.photo(64, 100, seed=7)
    for fmt in (Format.BC7, Format.ASTC_8x6, Format.ETC2_R8G8B8):
        q = make_params(fmt, Type.UNorm, 1)
        whole = gpu_ctx.encode([big], q)[0]
        parts = [shard.encode_rows_sharded(big, q, gpu_ctx.encode, r, 3, gather=False) for r in range(3)]
        assert np.array_equal(np.concatenate(parts), whole), fmt


def test_device_path_waits_for_torch_producers(gpu_ctx):
    """torch's legacy default stream has handle 0 -- "the context's own stream" to the C ABI, a
    non-blocking stream that does not order itself behind torch's work.  The device helpers finish
    the producers first: a surface still being written by queued torch kernels encodes to the same
    bytes as one copied to the host and back."""
    dev = torch.device("cuda", 0)
    p = make_params(Format.BC1_RGB, Type.UNorm, 0)             # a fast kernel: it would overtake the producer
    n = 2048
    for seed in range(3):
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        x = torch.rand((n, n, 4), generator=g, device=dev)
        for _ in range(6):                                      # a queue of dependent elementwise kernels
            x = (x * 1.7 + 0.1).frac()
        img = (x * 255.0).round().to(torch.uint8).contiguous()
        got = shard.encode_rows_sharded_device(gpu_ctx, img, n, n, PixelType.RGBA8, p, 0, 1)
        ref = gpu_ctx.encode([img.cpu().numpy()], p)[0]
        assert np.array_equal(got.cpu().numpy(), ref), seed


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, backend, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        with Context(dev) as ctx:
            p = make_params(Format.BC7, Type.UNorm, 2)
            got = shard.encode_surfaces_sharded(_chain(), p, ctx.encode, rank, world)
            rows = shard.encode_rows_sharded(synth.photo(64, 100, seed=7), p, ctx.encode, rank, world)
        q.put((rank, [g.tobytes() for g in got], rows.tobytes()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_world2_equals_world1(gpu_ctx):
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    port = _free_port()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, 2, port, backend, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    results = [q.get(timeout=480) for _ in range(2)]
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    p = make_params(Format.BC7, Type.UNorm, 2)
    want = [o.tobytes() for o in gpu_ctx.encode(_chain(), p)]
    want_rows = gpu_ctx.encode([synth.photo(64, 100, seed=7)], p)[0].tobytes()
    for rank, got, rows in results:
        assert got == want and rows == want_rows, "rank %d" % rank


def _dev_worker(rank, world, port, backend, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from cuttlefish_amd import PixelType
        with Context(dev) as ctx:
            p = make_params(Format.ASTC_6x6, Type.UNorm, 1)
            img = synth.photo(70, 100, seed=7)
            full = torch.from_numpy(img).cuda().reshape(-1) if rank == 0 else None
            tm = {}
            rows = shard.encode_rows_sharded_device(ctx, full, 70, 100, PixelType.RGBA8, p, rank, world,
                                                    src=0, dst=0, timings=tm)
            assert rows.is_cuda and set(tm) == {"scatter_s", "encode_s", "gather_s"}
            p7 = make_params(Format.BC7, Type.UNorm, 2)
            surfs = [{"width": im.shape[1], "height": im.shape[0], "pixel_type": PixelType.RGBA8,
                      "pixels": torch.from_numpy(im).cuda()} for im in _chain()]
            outs = shard.encode_surfaces_sharded_device(ctx, surfs, p7, rank, world, dst=None)
            assert all(o.is_cuda for o in outs)
            q.put((rank, rows.cpu().numpy().tobytes(), [o.cpu().numpy().tobytes() for o in outs]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_world2_device_buffer_path_equals_world1(gpu_ctx):
    """scatter -> encode_device -> gather with device tensors on both ends (RCCL when two devices
    are visible; on one device the ranks share it and gloo carries the bytes)."""
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    port = _free_port()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_dev_worker, args=(r, 2, port, backend, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    results = dict((r[0], r) for r in [q.get(timeout=480) for _ in range(2)])
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    pa = make_params(Format.ASTC_6x6, Type.UNorm, 1)
    want_rows = gpu_ctx.encode([synth.photo(70, 100, seed=7)], pa)[0].tobytes()
    want = [o.tobytes() for o in gpu_ctx.encode(_chain(), make_params(Format.BC7, Type.UNorm, 2))]
    assert results[0][1] == want_rows                    # rank 0 holds the whole payload
    assert results[1][1] == want_rows[8*12*16:]          # rank 1 keeps its own range: block rows 8..16 of 17, 12 blocks each
    for r in (0, 1):
        assert results[r][2] == want
