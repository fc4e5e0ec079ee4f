"""GPU parity for ETC1 / ETC2 RGB / RGBA1 / RGBA8 / EAC R11 / RG11: byte-exact vs the oracle."""
import numpy as np
import pytest

import oracle_lib as O
from cuttlefish_amd import ColorSpace, Context, Format, Texture, Type, make_params, synth

pytestmark = pytest.mark.gpu

RGBF = [Format.ETC1, Format.ETC2_R8G8B8, Format.ETC2_R8G8B8A1, Format.ETC2_R8G8B8A8]
EAC = [Format.EAC_R11, Format.EAC_R11G11]


def _gpu(ctx, img, fmt, typ=Type.UNorm, quality=2, **kw):
    return ctx.encode([img], make_params(fmt, typ, quality, **kw))[0]


def _check(ref, got, bs):
    bad = np.flatnonzero((ref.reshape(-1, bs) != got.reshape(-1, bs)).any(axis=1))
    assert bad.size == 0, "blocks differ: %s" % bad[:10]


@pytest.mark.parametrize("fmt", RGBF)
@pytest.mark.parametrize("quality", [0, 2, 4])
def test_rgb_family_bit_exact(gpu_ctx, fmt, quality):
    img = synth.photo(64, 48, seed=50 + quality)
    ref = O.encode(img, int(fmt), quality=quality, threads=8)
    _check(ref, _gpu(gpu_ctx, img, fmt, quality=quality), 16 if fmt == Format.ETC2_R8G8B8A8 else 8)


@pytest.mark.parametrize("fmt", EAC)
@pytest.mark.parametrize("typ", [Type.UNorm, Type.SNorm])
def test_eac_bit_exact_float_and_u8(gpu_ctx, fmt, typ):
    rng = np.random.default_rng(9)
    f = (rng.random((24, 28, 4)).astype(np.float32) * 2.4 - 1.2)
    ref = O.encode(f, int(fmt), typ=int(typ), quality=2, threads=8)
    _check(ref, _gpu(gpu_ctx, f, fmt, typ), 16 if fmt == Format.EAC_R11G11 else 8)
    u8 = synth.photo(28, 24, seed=8)
    ref = O.encode(u8, int(fmt), typ=int(typ), quality=1, threads=8)
    _check(ref, _gpu(gpu_ctx, u8, fmt, typ, quality=1), 16 if fmt == Format.EAC_R11G11 else 8)


@pytest.mark.parametrize("fmt", RGBF)
def test_noise_float_and_srgb_metric(gpu_ctx, fmt):
    rng = np.random.default_rng(int(fmt))
    f = (rng.random((20, 36, 4)).astype(np.float32) * 1.3 - 0.15)
    ref = O.encode(f, int(fmt), quality=2, threads=8)
    assert np.array_equal(ref, _gpu(gpu_ctx, f, fmt))
    img = synth.photo(32, 32, seed=3)
    if fmt != Format.ETC1:
        ref = O.encode(img, int(fmt), quality=2, threads=8, color_space=1)
        assert np.array_equal(ref, _gpu(gpu_ctx, img, fmt, color_space=ColorSpace.sRGB))


@pytest.mark.parametrize("w,h", [(1, 1), (6, 6), (17, 9), (66, 7)])
def test_partial_blocks(gpu_ctx, w, h):
    img = synth.photo(w, h, seed=w * 3 + h)
    for fmt in (Format.ETC2_R8G8B8, Format.ETC2_R8G8B8A8, Format.EAC_R11):
        ref = O.encode(img, int(fmt), quality=1, threads=4)
        assert np.array_equal(ref, _gpu(gpu_ctx, img, fmt, quality=1))


def test_texture_convert_contracts(gpu_ctx):
    img = np.zeros((16, 16, 4), np.float32)
    img[..., 3] = 1.0
    for fmt, typ, size in [(Format.ETC1, Type.UNorm, 128), (Format.ETC2_R8G8B8A8, Type.UNorm, 256),
                           (Format.EAC_R11G11, Type.SNorm, 256)]:
        t = Texture(16, 16)
        assert t.set_image(img)
        assert t.convert(fmt, typ) and t.data_size() == size
    t = Texture(16, 16)
    t.set_image(img)
    assert not t.convert(Format.ETC1, Type.SNorm)           # Converter.cpp:414-422
    s = Texture(16, 16, color_space=ColorSpace.sRGB)
    s.set_image(img)
    assert not s.convert(Format.ETC1, Type.UNorm)           # ETC1 has no sRGB variant (Texture.cpp:421-465)


def test_full_size_properties_etc2_2048(gpu_ctx):
    img = synth.photo(2048, 2048, seed=1)
    a = _gpu(gpu_ctx, img, Format.ETC2_R8G8B8A8, quality=2)
    assert a.nbytes == 512 * 512 * 16
    assert np.array_equal(a, _gpu(gpu_ctx, img, Format.ETC2_R8G8B8A8, quality=2))
    dec = O.decode_etc(a, int(Format.ETC2_R8G8B8A8), 2048, 2048)
    assert synth.psnr(img, dec, slice(0, 3)) > 34.0
    strip = img[1024:1040]
    ref = O.encode(strip, int(Format.ETC2_R8G8B8A8), quality=2, threads=8)
    assert np.array_equal(ref, a.reshape(512, 512 * 16)[256:260].reshape(-1))


@pytest.mark.parametrize("quality", [0, 1, 2, 3])
def test_punch_through_blocks_with_th_modes_match_the_oracle(quality):
    rng = np.random.default_rng(11)
    img = synth.photo(96, 64, seed=9).copy()
    img[..., 3] = np.where(rng.random((64, 96)) < 0.25, 0, 255)
    img[:8, :8, 3] = 0                                   # fully transparent blocks
    img[8:16, :, 3] = 255                                # fully opaque block rows
    with Context(0) as ctx:
        got = ctx.encode([img], make_params(Format.ETC2_R8G8B8A1, Type.UNorm, quality))[0]
    assert np.array_equal(got, O.encode(img, int(Format.ETC2_R8G8B8A1), 0, quality=quality, threads=8))


def test_a1_srgb_partial_blocks(gpu_ctx):
    """Round 5: the 5-wave build of the RGB8A1 kernel (heavy register spilling) returned wrong blocks for PARTIAL blocks
    (texels outside the image carry no weight) of sRGB images at Normal and above -- found by tools/fuzz_all.sh, three
    of 300 cases.  The blocks of those cases, as images of their valid extent."""
    cases = [("1e764cff1e764cff616223ff616223ff", 4, 1), ("a7fc1dff547b62ffd53752ff9830a8ff", 4, 1),
             ("7e07dc9a38cfbf40be93e78762570c77", 1, 4)]
    for px, w, h in cases:
        img = np.ascontiguousarray(np.frombuffer(bytes.fromhex(px), np.uint8).reshape(h, w, 4))
        for fmt in (Format.ETC2_R8G8B8A1, Format.ETC2_R8G8B8, Format.ETC2_R8G8B8A8):
            for cs in (0, 1):
                for q in range(5):
                    ref = O.encode(img, int(fmt), quality=q, threads=1, color_space=cs)
                    got = gpu_ctx.encode([img], make_params(fmt, Type.UNorm, q, color_space=ColorSpace(cs)))[0]
                    assert np.array_equal(ref, got), (fmt.name, w, h, cs, q)
