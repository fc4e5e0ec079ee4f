"""cfhip_encode_multi: several contexts of ONE process share the surfaces of a call (the
reference's CLI / library are a single process).  On a one-GPU box the contexts live on the same
device, which exercises the assignment, the worker threads and the per-thread device selection;
with more devices visible each context gets its own."""
import numpy as np
import pytest

from cuttlefish_amd import Context, Format, Type, api, make_params, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fmt,quality,nctx", [(Format.BC7, 2, 3), (Format.ASTC_6x6, 1, 3), (Format.ETC2_R8G8B8A8, 2, 3),
                                              (Format.BC7, 1, 8), (Format.BC3, 2, 8)])
def test_multi_context_equals_single_context(gpu_ctx, fmt, quality, nctx):
    """nctx = 8: the context count of the 8-GPU node (here they share the visible devices): 9 surfaces over 8
    contexts, so one context gets two small ones and the assignment's tail is exercised."""
    params = make_params(fmt, Type.UNorm, quality)
    chain = [synth.photo(max(1, 96 >> i), max(1, 64 >> i), seed=200 + i) for i in range(7)]
    chain += [synth.photo(52, 36, seed=300), synth.photo(5, 3, seed=301)]
    want = gpu_ctx.encode(chain, params)
    ndev = api.device_count()
    others = [Context(d % ndev) for d in range(1, nctx)]
    try:
        got = gpu_ctx.encode_multi(others, chain, params)
    finally:
        for c in others:
            c.close()
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


def test_multi_reports_the_failure_of_any_share(gpu_ctx):
    params = make_params(Format.BC7, Type.UNorm, 2)            # refused: BC7 takes RGBA8 or RGBA32F pixels
    imgs = [(synth.photo(16, 16, seed=1).astype(np.float32)/255.0).astype(np.float16) for _ in range(3)]
    other = Context(0)
    try:
        with pytest.raises(api.CfhipError):
            gpu_ctx.encode_multi([other], imgs, params)
    finally:
        other.close()


@pytest.mark.parametrize("nctx", [3, 8])
@pytest.mark.parametrize("fmt,quality,w,h", [(Format.ASTC_6x6, 1, 1024, 1000), (Format.BC7, 2, 1024, 770),
                                              (Format.ETC2_R8G8B8, 1, 515, 1021), (Format.R8G8B8, 2, 333, 1027)])
def test_one_big_surface_is_row_split_over_the_contexts(gpu_ctx, fmt, quality, w, h, nctx):
    """A surface holding more than 1/n of the call's blocks is cut into block-row ranges, one per
    context (the reference parallelises INSIDE a surface, Converter.cpp:540-583): byte-identical
    to the one-context encode -- ragged bottom edge, ASTC block height 6, a 3-byte standard
    format (ranges in multiples of 4 rows) and a small companion surface that is not split."""
    params = make_params(fmt, Type.UNorm, quality)
    tile = synth.photo(256, 256, seed=77)
    big = np.ascontiguousarray(np.tile(tile, (h // 256 + 1, w // 256 + 1, 1))[:h, :w])
    big[::7, ::5, :3] ^= 0x15                        # break the tiling period
    imgs = [big, synth.photo(40, 24, seed=78)]
    want = gpu_ctx.encode(imgs, params)
    ndev = api.device_count()
    others = [Context(d % ndev) for d in range(1, nctx)]
    try:
        got = gpu_ctx.encode_multi(others, imgs, params)
        kernel_ms = [c.last_kernel_ms() for c in [gpu_ctx] + others]
    finally:
        for c in others:
            c.close()
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    # every context really encoded something
    assert all(ms > 0 for ms in kernel_ms), kernel_ms


def test_row_split_keeps_the_capacity_check(gpu_ctx):
    import ctypes
    params = make_params(Format.BC7, Type.UNorm, 0)
    img = np.zeros((512, 512, 4), np.uint8)
    other = Context(0)
    try:
        surf, outs, keep = gpu_ctx._host_surfaces([img], params)
        surf[0].out_capacity = outs[0].nbytes - 16
        arr = (ctypes.c_void_p * 2)(gpu_ctx._h, other._h)
        rc = gpu_ctx._lib.cfhip_encode_multi(arr, 2, surf, 1, ctypes.byref(params))
        assert rc == api.E_CAPACITY
    finally:
        other.close()


@pytest.mark.parametrize("nctx", [1, 3])
def test_release_hook_reports_every_surface_once(gpu_ctx, nctx):
    """cfhip_encode_multi_ex: `consumed(i)` fires exactly once per surface -- also for a surface that was cut
    into block-row ranges over the contexts (once, after its LAST range was read), for the strip-pipelined
    float surface and for the small surfaces uploaded as a group -- and the payloads are the plain call's.
    This is the hook through which HipConverter frees each source image as Converter.cpp:586 does."""
    import threading
    params = make_params(Format.BC7, Type.UNorm, 1)
    big = np.ascontiguousarray(np.tile(synth.photo(256, 256, seed=5), (5, 4, 1))[:1100, :1000])
    flt = (synth.photo(640, 512, seed=6).astype(np.float32)/255.0)[::-1]        # bottom-up view: pipelined path
    imgs = [big, flt] + [synth.photo(max(1, 64 >> i), max(1, 64 >> i), seed=7 + i) for i in range(7)]
    want = gpu_ctx.encode(imgs, params)
    seen, lock = [], threading.Lock()

    def consumed(i):
        with lock:
            seen.append(i)
    others = [Context(d % api.device_count()) for d in range(1, nctx)]
    try:
        got = gpu_ctx.encode_multi(others, imgs, params, consumed=consumed)
    finally:
        for c in others:
            c.close()
    assert sorted(seen) == list(range(len(imgs)))
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
