"""GPU parity for BC1/BC1A/BC2/BC3/BC4/BC5: HIP kernels (through the C-ABI) against the CPU
oracle on the same inputs -- byte-exact (integer path)."""
import numpy as np
import pytest

import oracle_lib as O
from cuttlefish_amd import ColorSpace, Format, Texture, Type, make_params, synth

pytestmark = pytest.mark.gpu

FORMATS = [Format.BC1_RGB, Format.BC1_RGBA, Format.BC2, Format.BC3, Format.BC4, Format.BC5]


def _gpu(ctx, img, fmt, typ=Type.UNorm, quality=2, **kw):
    return ctx.encode([img], make_params(fmt, typ, quality, **kw))[0]


def _assert_blocks_equal(ref, got, bs):
    assert ref.nbytes == got.nbytes
    bad = np.flatnonzero((ref.reshape(-1, bs) != got.reshape(-1, bs)).any(axis=1))
    assert bad.size == 0, "blocks differ: %s" % bad[:10]


@pytest.mark.parametrize("fmt", FORMATS)
@pytest.mark.parametrize("quality", [0, 1, 2, 3, 4])
def test_bit_exact_vs_oracle(gpu_ctx, fmt, quality):
    img = synth.photo(80, 48, seed=20 + quality)
    ref = O.encode(img, int(fmt), quality=quality, threads=8)
    got = _gpu(gpu_ctx, img, fmt, quality=quality)
    _assert_blocks_equal(ref, got, 8 if fmt in (Format.BC1_RGB, Format.BC1_RGBA, Format.BC4) else 16)


@pytest.mark.parametrize("fmt", FORMATS)
def test_bit_exact_random_noise(gpu_ctx, fmt):
    rng = np.random.default_rng(int(fmt))
    img = rng.integers(0, 256, (32, 48, 4), dtype=np.uint8)
    ref = O.encode(img, int(fmt), quality=2, threads=8)
    assert np.array_equal(ref, _gpu(gpu_ctx, img, fmt))


@pytest.mark.parametrize("fmt", [Format.BC4, Format.BC5])
def test_snorm_bit_exact_float_and_u8_sources(gpu_ctx, fmt):
    rng = np.random.default_rng(5)
    f = (rng.random((24, 36, 4)).astype(np.float32) * 2.4 - 1.2)
    ref = O.encode(f, int(fmt), typ=1, quality=2, threads=8)
    assert np.array_equal(ref, _gpu(gpu_ctx, f, fmt, Type.SNorm))
    u8 = synth.photo(36, 24, seed=6)
    ref = O.encode(u8, int(fmt), typ=1, quality=2, threads=8)
    assert np.array_equal(ref, _gpu(gpu_ctx, u8, fmt, Type.SNorm))


@pytest.mark.parametrize("fmt", FORMATS)
def test_float_source_matches(gpu_ctx, fmt):
    rng = np.random.default_rng(17)
    f = (rng.random((20, 28, 4)).astype(np.float32) * 1.3 - 0.15)
    ref = O.encode(f, int(fmt), quality=2, threads=8)
    assert np.array_equal(ref, _gpu(gpu_ctx, f, fmt))


def test_bc1a_weights_and_mask_in_punch_through_blocks(gpu_ctx):
    img = synth.photo(64, 64, seed=4, alpha=True)
    for kw_o, kw_g in [({"color_space": 1}, {"color_space": ColorSpace.sRGB}),
                       ({"mask": (1, 0, 1, 1)}, {"color_mask": (1, 0, 1, 1)})]:
        ref = O.encode(img, int(Format.BC1_RGBA), quality=2, threads=8, **kw_o)
        assert np.array_equal(ref, _gpu(gpu_ctx, img, Format.BC1_RGBA, **kw_g))


@pytest.mark.parametrize("w,h", [(1, 1), (5, 3), (17, 9), (67, 6)])
def test_ragged_sizes(gpu_ctx, w, h):
    img = synth.photo(w, h, seed=w + 13 * h)
    for fmt in (Format.BC1_RGB, Format.BC3, Format.BC5):
        ref = O.encode(img, int(fmt), quality=1, threads=4)
        assert np.array_equal(ref, _gpu(gpu_ctx, img, fmt, quality=1))


def test_config1_bc1_512_gradient(gpu_ctx):
    """BASELINE config 1: BC1 of the reference's 512x512 gradient (TextureTest.cpp:53-61)
    through the Texture.convert mirror."""
    img = synth.gradient(512, 512, np.float32)
    t = Texture(512, 512)
    assert t.set_image(img)
    assert t.convert(Format.BC1_RGB, Type.UNorm)
    assert t.data_size() == 128 * 128 * 8
    ref = O.encode(img, int(Format.BC1_RGB), quality=2, threads=8)
    assert np.array_equal(ref, t.data())
    dec = O.decode(t.data(), int(Format.BC1_RGB), 512, 512)
    u8 = synth.gradient(512, 512, np.uint8)
    assert synth.psnr(u8, dec, slice(0, 3)) > 40.0


def test_full_size_properties_bc3_4096(gpu_ctx):
    img = synth.photo(4096, 4096, seed=1)
    a = _gpu(gpu_ctx, img, Format.BC3, quality=2)
    assert a.nbytes == 1024 * 1024 * 16
    assert np.array_equal(a, _gpu(gpu_ctx, img, Format.BC3, quality=2))
    dec = O.decode(a, int(Format.BC3), 4096, 4096)
    assert synth.psnr(img, dec, slice(0, 3)) > 40.0
    strip = img[2048:2064]
    ref = O.encode(strip, int(Format.BC3), quality=2, threads=8)
    assert np.array_equal(ref, a.reshape(1024, 1024 * 16)[512:516].reshape(-1))
