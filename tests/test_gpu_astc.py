"""GPU parity for ASTC (restricted LDR subset): byte-exact vs the oracle, all 14 footprints."""
import numpy as np
import pytest

import oracle_lib as O
from cuttlefish_amd import Alpha, ColorSpace, Format, Texture, Type, make_params, synth

pytestmark = pytest.mark.gpu
FORMATS = [Format(v) for v in range(43, 57)]


def _gpu(ctx, img, fmt, quality=2, **kw):
    return ctx.encode([img], make_params(fmt, Type.UNorm, quality, **kw))[0]


@pytest.mark.parametrize("fmt", FORMATS)
def test_bit_exact_all_footprints(gpu_ctx, fmt):
    img = synth.photo(70, 50, seed=int(fmt))
    ref = O.encode(img, int(fmt), quality=2, threads=8)
    got = _gpu(gpu_ctx, img, fmt)
    bad = np.flatnonzero((ref.reshape(-1, 16) != got.reshape(-1, 16)).any(axis=1))
    assert bad.size == 0, "blocks differ: %s" % bad[:10]


@pytest.mark.parametrize("fmt", [Format.ASTC_6x6, Format.ASTC_8x8, Format.ASTC_10x6, Format.ASTC_10x10, Format.ASTC_12x12])
@pytest.mark.parametrize("quality", [2, 3, 4])
def test_bit_exact_two_channel_gradients(gpu_ctx, fmt, quality):
    """round 6: the content the coarse grids and the least-squares step are for (tests/test_oracle_astc.py:
    test_two_channel_gradients_take_a_coarse_dual_plane_grid) -- the kernel emits the oracle's bytes on it"""
    import real_lib as R
    bw, bh = {Format.ASTC_6x6: (6, 6), Format.ASTC_8x8: (8, 8), Format.ASTC_10x6: (10, 6), Format.ASTC_10x10: (10, 10),
              Format.ASTC_12x12: (12, 12)}[fmt]
    img = np.ascontiguousarray(np.tile(R.two_channel_gradients(bw, bh), (3, 1, 1)))
    ref = O.encode(img, int(fmt), quality=quality, threads=8)
    assert np.array_equal(ref, _gpu(gpu_ctx, img, fmt, quality))


@pytest.mark.parametrize("quality", [0, 1, 3, 4])
def test_bit_exact_quality_ladder_6x6(gpu_ctx, quality):
    img = synth.photo(66, 42, seed=90 + quality)
    ref = O.encode(img, 47, quality=quality, threads=8)
    assert np.array_equal(ref, _gpu(gpu_ctx, img, Format.ASTC_6x6, quality))


@pytest.mark.parametrize("fmt", [Format.ASTC_4x4, Format.ASTC_5x5, Format.ASTC_6x5, Format.ASTC_6x6, Format.ASTC_10x5,
                                 Format.ASTC_8x8, Format.ASTC_12x10])
@pytest.mark.parametrize("quality", [2, 3, 4])
def test_refinement_quads_every_metric_and_ragged_rows(gpu_ctx, fmt, quality):
    """Round 5: the refinement rounds run on the group's best results, four lanes per result.  Footprints whose texel
    count is not a multiple of four (5x5, 6x5, 10x5), block rows with an odd number of blocks (the lone block of a
    level that pairs blocks), images with and without alpha, the perceptual metric of sRGB images and the three
    alpha types (AstcConverter.cpp:163-172: the quad then walks the channel-weighted forms of reprojection and exact
    error) -- byte for byte against the oracle."""
    import ctypes
    cbw, cbh = ctypes.c_int(), ctypes.c_int()
    O.lib().cfo_astc_footprint(int(fmt), ctypes.byref(cbw), ctypes.byref(cbh))
    bw, bh = cbw.value, cbh.value
    w, h = 9*bw - 2, 3*bh + 1                   # nine blocks per row (the wave's ninth is alone), ragged edges
    img = synth.photo(w, h, seed=300 + int(fmt) + quality)
    opaque = img.copy()
    opaque[..., 3] = 255
    cases = [(img, dict(color_space=1)), (opaque, dict(color_space=1, alpha=0)), (img, dict(alpha=2)), (opaque, dict())]
    for im, kw in cases:
        ref = O.encode(im, int(fmt), quality=quality, threads=8, **kw)
        gkw = dict(kw)
        if "alpha" in gkw:
            gkw["alpha"] = Alpha(gkw["alpha"])
        if "color_space" in gkw:
            gkw["color_space"] = ColorSpace(gkw["color_space"])
        got = _gpu(gpu_ctx, im, fmt, quality, **gkw)
        bad = np.flatnonzero((ref.reshape(-1, 16) != got.reshape(-1, 16)).any(axis=1))
        assert bad.size == 0, (kw, "blocks differ: %s" % bad[:10])


def test_float_source_noise_swizzle_and_solid(gpu_ctx):
    rng = np.random.default_rng(4)
    f = (rng.random((31, 45, 4)).astype(np.float32) * 1.3 - 0.15)
    ref = O.encode(f, 47, quality=2, threads=8)
    assert np.array_equal(ref, _gpu(gpu_ctx, f, Format.ASTC_6x6))
    img = synth.photo(48, 48, seed=6)
    ref = O.encode(img, 45, quality=2, threads=8, mask=(1, 0, 1, 1))
    assert np.array_equal(ref, _gpu(gpu_ctx, img, Format.ASTC_5x5, color_mask=(1, 0, 1, 1)))
    ref = O.encode(img, 45, quality=2, threads=8, alpha=0)
    assert np.array_equal(ref, _gpu(gpu_ctx, img, Format.ASTC_5x5, alpha=Alpha.None_))
    solid = np.full((24, 24, 4), 77, np.uint8)
    ref = O.encode(solid, 50, quality=2)
    assert np.array_equal(ref, _gpu(gpu_ctx, solid, Format.ASTC_8x8))


def test_texture_convert_size_contract_and_type_legality(gpu_ctx):
    img = np.zeros((16, 16, 4), np.float32)
    img[..., 3] = 1.0
    t = Texture(16, 16)
    assert t.set_image(img)
    assert t.convert(Format.ASTC_6x6, Type.UNorm)          # 3 x 3 blocks (ASTC_6x6 is absent from
    assert t.data_size() == 9 * 16                          # the reference's own list, TextureTest.cpp:941-954)
    t = Texture(16, 16)
    t.set_image(img)
    assert not t.convert(Format.ASTC_6x6, Type.SNorm)      # createConverter -> nullptr
    assert t.convert(Format.ASTC_6x6, Type.UFloat)         # HDR profile (tests/test_gpu_astc_hdr.py)
    assert t.data_size() == 9 * 16


def test_config3_full_size_properties_4096(gpu_ctx):
    """BASELINE config 3: ASTC 6x6 'thorough' (Quality::High) on 4096x4096 RGBA8."""
    img = synth.photo(4096, 4096, seed=1)
    a = _gpu(gpu_ctx, img, Format.ASTC_6x6, 3)
    assert a.nbytes == 683 * 683 * 16
    assert np.array_equal(a, _gpu(gpu_ctx, img, Format.ASTC_6x6, 3))
    dec, bad = O.decode_astc(a, 47, 4096, 4096)
    assert bad == 0 and synth.psnr(img, dec, slice(0, 3)) > 38.0   # measured 39.99 (41.4 outside the alpha band)
    strip = img[2046:2058]                                  # block rows 341..342
    ref = O.encode(strip, 47, quality=3, threads=8)
    assert np.array_equal(ref, a.reshape(683, 683 * 16)[341:343].reshape(-1))
