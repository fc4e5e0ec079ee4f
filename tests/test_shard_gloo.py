"""N>1 path on CPU: world_size-2 and world_size-8 gloo processes shard surfaces / block rows and gather the
payload; the result must be byte-identical to the single-rank result.  The encoder injected
here is the CPU oracle (tests may use it); on GPUs the same code runs with Context.encode."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from cuttlefish_amd import Format, Type, make_params, shard, synth


def _oracle_encode(images, params):
    return [O.encode(np.ascontiguousarray(im), int(params.format), typ=int(params.type),
                     quality=int(params.quality)) for im in images]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params = make_params(Format.BC3, Type.UNorm, 1)
        chain = [synth.photo(32 >> i, 32 >> i, seed=60 + i) for i in range(6)]
        chain += [synth.photo(20, 12, seed=70), synth.photo(8, 40, seed=71)]
        got = shard.encode_surfaces_sharded(chain, params, _oracle_encode, rank, world)
        img = synth.photo(40, 52, seed=80)
        rows = shard.encode_rows_sharded(img, params, _oracle_encode, rank, world)
        # a footprint whose block height is not 4: shards must start on ASTC block rows
        ap = make_params(Format.ASTC_6x6, Type.UNorm, 0)
        aimg = synth.photo(30, 44, seed=81)
        arows = shard.encode_rows_sharded(aimg, ap, _oracle_encode, rank, world)
        asurf = shard.encode_surfaces_sharded([aimg, aimg[:13, :7]], ap, _oracle_encode, rank, world)
        q.put((rank, [g.tobytes() for g in got], rows.tobytes(), arows.tobytes(),
               [g.tobytes() for g in asurf]))
    finally:
        dist.destroy_process_group()


def test_assign_surfaces_is_balanced_and_deterministic():
    counts = [shard.block_count(2048 >> i, 2048 >> i) for i in range(12)] * 4
    plan = shard.assign_surfaces(counts, 8)
    assert sorted(i for p in plan for i in p) == list(range(len(counts)))
    loads = [sum(counts[i] for i in p) for p in plan]
    assert max(loads) - min(loads) <= max(counts)
    assert plan == shard.assign_surfaces(counts, 8)
    assert shard.assign_surfaces([5, 1], 4) == [[0], [1], [], []]


@pytest.mark.timeout(300)
def test_world2_gloo_matches_single_rank():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0

    params = make_params(Format.BC3, Type.UNorm, 1)
    chain = [synth.photo(32 >> i, 32 >> i, seed=60 + i) for i in range(6)]
    chain += [synth.photo(20, 12, seed=70), synth.photo(8, 40, seed=71)]
    want = [o.tobytes() for o in _oracle_encode(chain, params)]
    img = synth.photo(40, 52, seed=80)
    want_rows = _oracle_encode([img], params)[0].tobytes()
    ap = make_params(Format.ASTC_6x6, Type.UNorm, 0)
    aimg = synth.photo(30, 44, seed=81)
    want_a = [o.tobytes() for o in _oracle_encode([aimg, aimg[:13, :7]], ap)]
    assert len(want_a[0]) == 5*8*16                     # ceil(30/6) x ceil(44/6) blocks
    for rank, got, rows, arows, asurf in results:
        assert got == want, "rank %d surfaces differ" % rank
        assert rows == want_rows, "rank %d row shards differ" % rank
        assert arows == want_a[0], "rank %d ASTC 6x6 row shards differ" % rank
        assert asurf == want_a, "rank %d ASTC surfaces differ" % rank


def test_row_shards_follow_the_format_block_height():
    """h = 24 with ASTC 6x6 is 4 block rows (not 6): every shard boundary is a multiple of 6."""
    from cuttlefish_amd import api
    seen = []

    def spy(images, params):
        seen.append(images[0].shape[0])
        return _oracle_encode(images, params)
    img = synth.photo(12, 24, seed=5)
    ap = make_params(Format.ASTC_6x6, Type.UNorm, 0)
    whole = _oracle_encode([img], ap)[0]
    parts = [shard.encode_rows_sharded(img, ap, spy, r, 4, gather=False) for r in range(4)]
    assert seen == [6, 6, 6, 6]
    assert np.array_equal(np.concatenate(parts), whole)
    assert shard.block_count(12, 24, *api.query(Format.ASTC_6x6, Type.UNorm)[:2]) == 2*4


def _scatter_worker(rank, world, port, q):
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params = make_params(Format.ASTC_6x6, Type.UNorm, 0)
        h, w = 44, 30
        ranges = shard.row_ranges(h, 6, world)
        sizes = [(y1 - y0)*w*4 for (y0, y1, _, _) in ranges]
        parts = None
        if rank == 1:                    # the surface lives on rank 1: scatter from there
            flat = torch.from_numpy(synth.photo(w, h, seed=81).reshape(-1).copy())
            parts = [flat[y0*w*4:y1*w*4] for (y0, y1, _, _) in ranges]
        mine = shard.scatter(parts, sizes, rank, world, src=1, device=torch.device("cpu"))
        y0, y1, a, b = ranges[rank]
        rows = mine.numpy().reshape(y1 - y0, w, 4)
        local = torch.from_numpy(_oracle_encode([rows], params)[0].reshape(-1).copy())
        out_sizes = [(rb - ra)*5*16 for (_, _, ra, rb) in ranges]
        got = shard.exchange(local, out_sizes, rank, world, dst=0)
        q.put((rank, ranges, None if rank else b"".join(p.numpy().tobytes() for p in got)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_scatter_encode_gather_equals_single_rank():
    """The device-path building blocks on CPU tensors: exact-size scatter of block-row ranges from
    the rank holding the surface, encode, exact-size gather to rank 0."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_scatter_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict((r[0], r) for r in [q.get(timeout=240) for _ in range(world)])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ap = make_params(Format.ASTC_6x6, Type.UNorm, 0)
    want = _oracle_encode([synth.photo(30, 44, seed=81)], ap)[0].tobytes()
    assert results[0][2] == want
    assert results[1][2] is None
    # 8 block rows of 6 scanlines (the last one ragged: 44 = 7*6 + 2) split 4 + 4
    assert results[0][1] == [(0, 24, 0, 4), (24, 44, 4, 8)]


def _worker8(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params = make_params(Format.BC1_RGB, Type.UNorm, 0)
        # 3 chains of 5 levels (15 surfaces over 8 ranks: some ranks get one tiny surface) ...
        chain = [synth.photo(32 >> i, 32 >> i, seed=90 + 10*t + i) for t in range(3) for i in range(5)]
        got = shard.encode_surfaces_sharded(chain, params, _oracle_encode, rank, world)
        # ... fewer surfaces than ranks: zero-byte shares in the exchange
        few = shard.encode_surfaces_sharded(chain[:3], params, _oracle_encode, rank, world)
        # one surface of 13 block rows over 8 ranks, and one of 3 block rows (fewer rows than ranks:
        # five ranks own nothing and send nothing)
        tall = synth.photo(24, 52, seed=120)
        rows = shard.encode_rows_sharded(tall, params, _oracle_encode, rank, world)
        short = synth.photo(40, 10, seed=121)
        srows = shard.encode_rows_sharded(short, params, _oracle_encode, rank, world)
        q.put((rank, [g.tobytes() for g in got], [g.tobytes() for g in few], rows.tobytes(), srows.tobytes()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_world8_gloo_matches_single_rank():
    """The world size the 8-GPU node runs, on CPU: both sharding functions, ranks with empty
    shares, a surface with fewer block rows than ranks."""
    world = 8
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=500) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    params = make_params(Format.BC1_RGB, Type.UNorm, 0)
    chain = [synth.photo(32 >> i, 32 >> i, seed=90 + 10*t + i) for t in range(3) for i in range(5)]
    want = [o.tobytes() for o in _oracle_encode(chain, params)]
    want_rows = _oracle_encode([synth.photo(24, 52, seed=120)], params)[0].tobytes()
    want_short = _oracle_encode([synth.photo(40, 10, seed=121)], params)[0].tobytes()
    assert sorted(r[0] for r in results) == list(range(world))
    for rank, got, few, rows, srows in results:
        assert got == want, "rank %d surfaces differ" % rank
        assert few == want[:3], "rank %d: fewer surfaces than ranks" % rank
        assert rows == want_rows, "rank %d row shards differ" % rank
        assert srows == want_short, "rank %d: fewer block rows than ranks" % rank
    # the plan itself: 3 block rows over 8 ranks leave five ranks empty, nothing is lost or doubled
    ranges = shard.row_ranges(10, 4, world)
    assert sum(b - a for (_, _, a, b) in ranges) == 3 and sum(1 for (_, _, a, b) in ranges if b > a) == 3
    assert [y1 - y0 for (y0, y1, _, _) in ranges if y1 > y0] == [4, 4, 2]
