"""Host-buffer entry point cfhip_encode: the pipelined strip path must
give the bytes of the plain path / the oracle for every source layout it accepts."""
import numpy as np
import pytest

import oracle_lib as O
from cuttlefish_amd import Context, Format, Type, make_params, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    with Context(0) as c:
        yield c


def test_large_float_surface_goes_through_strips_and_matches_oracle(ctx):
    img = synth.photo(1280, 1030, seed=21)            # 1030 rows: partial last block row in the last strip
    f = img.astype(np.float32)/np.float32(255.0)      # 21 MB of floats: pipelined, several strips
    got = ctx.encode([f], make_params(Format.BC7, Type.UNorm, 1))[0]
    assert np.array_equal(got, O.encode(img, int(Format.BC7), 0, quality=1, threads=16))


@pytest.mark.parametrize("fmt", [Format.BC1_RGB, Format.BC3, Format.BC7, Format.ETC2_R8G8B8A8, Format.ASTC_6x6])
def test_float_sources_are_quantised_on_the_host_like_toColorBlock(ctx, fmt):
    rng = np.random.default_rng(int(fmt))
    f = (rng.random((520, 1024, 4)).astype(np.float32)*1.2 - 0.1)     # out-of-range values clamp
    f[0, 0] = [np.nan, -0.0, 1e30, 0.5/255.0]                        # NaN -> 0, tie rounds away from zero
    u8 = np.round(np.clip(np.nan_to_num(f.astype(np.float64), nan=0.0), 0.0, 1.0)*np.float32(255.0))
    u8 = np.floor(np.clip(np.nan_to_num(f, nan=0.0), 0, 1).astype(np.float32)*np.float32(255) + np.float32(0.5)).astype(np.uint8)
    p = make_params(fmt, Type.UNorm, 1)
    got = ctx.encode([f], p)[0]                       # 8.5 MB of floats: pipelined + host quantisation
    ref = ctx.encode([u8], p)[0]                      # 2 MB of bytes: plain path
    assert np.array_equal(got, ref)


def test_bottom_up_image_negative_pitch(ctx):
    img = synth.photo(256, 192, seed=5)
    flipped_storage = np.ascontiguousarray(img[::-1])
    view = flipped_storage[::-1]                      # top-down view of bottom-up storage
    assert view.strides[0] < 0 and np.array_equal(view, img)
    p = make_params(Format.BC3, Type.UNorm, 2)
    assert np.array_equal(ctx.encode([view], p)[0], ctx.encode([img], p)[0])


def test_float_formats_keep_float_pixels_in_the_pipeline(ctx):
    hdr = synth.hdr_probe(1024, 512, seed=3).astype(np.float32)      # 8 MB RGBA32F -> BC6H
    p = make_params(Format.BC6H, Type.UFloat, 1)
    got = ctx.encode([hdr], p)[0]
    assert np.array_equal(got, O.encode(hdr, int(Format.BC6H), int(Type.UFloat), quality=1, threads=16))


def test_16k_surface_lands_through_a_small_pinned_ring():
    """Round-5 VERDICT item 8: the payload used to land in ONE pinned buffer as large as the largest payload the context
    had ever produced (256 MB after a 16k x 16k BC7 surface, never shrunk).  It now lands in a ring of four pinned strips:
    a 16384 x 16384 RGBA8 surface (1 GB of texels, 256 MB of BC7) goes through cfhip_encode with < 32 MB of page-locked
    memory in the context, and the bytes are those of the tile it is made of."""
    tile = synth.photo(1024, 1024, seed=9)
    # bottom-up storage, top-down view (an Image's scanlines, Image.cpp:340-343): the layout that takes the strip pipeline
    # for RGBA8 too (a plain top-down RGBA8 surface is one pageable upload and never touches pinned memory)
    big = np.ascontiguousarray(np.tile(tile[::-1], (16, 16, 1)))[::-1]
    assert big.strides[0] < 0 and np.array_equal(big[:1024, :1024], tile)
    p = make_params(Format.BC7, Type.UNorm, 0)
    with Context(0) as c:
        small = c.encode([tile], p)[0].reshape(256, 256, 16)
        got = c.encode([big], p)[0]
        assert 0 < c.pinned_bytes() <= 32 << 20, c.pinned_bytes()
        assert np.array_equal(got.reshape(4096, 4096, 16), np.tile(small, (16, 16, 1)))
        # a second, small pipelined surface on the same context: the ring is reused, nothing grows
        before = c.pinned_bytes()
        f = tile.astype(np.float32)/np.float32(255.0)
        assert np.array_equal(c.encode([f], p)[0].reshape(256, 256, 16), small)
        assert c.pinned_bytes() == before


def test_payload_goes_straight_to_the_callers_buffer_when_no_pinned_ring_can_be_had(monkeypatch):
    """a failed hipHostMalloc of the landing ring is not an error: the downloads target the caller's pageable buffer
    (CFHIP_NO_PINNED_OUT makes the context behave as if the allocation had failed)"""
    img = synth.photo(1280, 1030, seed=21)
    f = img.astype(np.float32)/np.float32(255.0)
    p = make_params(Format.BC7, Type.UNorm, 1)
    with Context(0) as c:
        want = c.encode([f], p)[0]
    monkeypatch.setenv("CFHIP_NO_PINNED_OUT", "1")
    with Context(0) as c:
        got = c.encode([f], p)[0]
        assert c.pinned_bytes() <= 3*(5 << 20)          # the source strip slots only
    assert np.array_equal(got, want)


def test_landing_ring_under_concurrent_contexts():
    """three host threads, each with its own context, push pipelined surfaces of different sizes and layouts through their
    landing rings at the same time and repeatedly (ring reuse across calls, slots larger and smaller than the last call's):
    every payload equals the plain top-down RGBA8 path's"""
    import threading
    rng = np.random.default_rng(5)
    cases = []
    for (w, h, fmt) in ((1280, 1030, Format.BC7), (2048, 516, Format.BC3), (777, 333, Format.ETC2_R8G8B8), (4096, 2052, Format.BC1_RGB)):
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        cases.append((img, fmt))
    with Context(0) as c:
        want = [c.encode([img], make_params(fmt, Type.UNorm, 0))[0] for img, fmt in cases]
    errors = []

    def worker(k):
        try:
            with Context(0) as c:
                for it in range(3):
                    for j in range(len(cases)):
                        img, fmt = cases[(j + k) % len(cases)]
                        p = make_params(fmt, Type.UNorm, 0)
                        if (it + j + k) & 1:
                            src = (img.astype(np.float32)/np.float32(255.0))            # float source: host quantiser + strips
                        else:
                            src = np.ascontiguousarray(img[::-1])[::-1]                  # bottom-up storage: strips
                        got = c.encode([src], p)[0]
                        if not np.array_equal(got, want[(j + k) % len(cases)]):
                            errors.append((k, it, j))
                assert c.pinned_bytes() <= 32 << 20
        except Exception as e:              # noqa: BLE001
            errors.append((k, repr(e)))
    ts = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
