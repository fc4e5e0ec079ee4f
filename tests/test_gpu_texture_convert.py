"""The reference's own TextureConvertTest / TextureConvertSpecialTest (lib/test/TextureTest.cpp:
823-985) run through the Texture mirror on the GPU: a 16x16 (0,0,0,1) RGBAF image converts to
every (format, type) of the reference's instantiation lists and data_size() is
blocksX*blocksY*blockSize; what createConverter rejects makes convert() return False; the sRGB
gate of TextureTest.cpp:815-820.  ASTC UFloat (the HDR profiles) is encoded with HDR endpoint modes."""
import numpy as np
import pytest

import oracle_lib as O
from test_oracle_stdpack import ALL_PAIRS
from cuttlefish_amd import ColorSpace, Format, Texture, Type, query

pytestmark = pytest.mark.gpu

U, S, UI, I, UF, F = Type.UNorm, Type.SNorm, Type.UInt, Type.Int, Type.UFloat, Type.Float
COMPRESSED = [(Format.BC1_RGB, [U]), (Format.BC1_RGBA, [U]), (Format.BC2, [U]), (Format.BC3, [U]),
              (Format.BC4, [U, S]), (Format.BC5, [U, S]), (Format.BC6H, [UF, F]), (Format.BC7, [U]),
              (Format.ETC1, [U]), (Format.ETC2_R8G8B8, [U]), (Format.ETC2_R8G8B8A1, [U]),
              (Format.ETC2_R8G8B8A8, [U]), (Format.EAC_R11, [U, S]), (Format.EAC_R11G11, [U, S])] + \
             [(Format(v), [U, UF]) for v in range(int(Format.ASTC_4x4), int(Format.ASTC_12x12) + 1)]
CASES = [(Format(f), Type(t)) for f, t in ALL_PAIRS] + [(f, t) for f, ts in COMPRESSED for t in ts]


def black():
    img = np.zeros((16, 16, 4), np.float32)
    img[..., 3] = 1.0
    return img


@pytest.mark.parametrize("fmt,typ", CASES, ids=lambda v: getattr(v, "name", str(v)))
def test_convert_size_contract(fmt, typ):
    t = Texture(16, 16)
    assert t.set_image(black())
    assert t.convert(fmt, typ)
    bw, bh, bs = query(fmt, typ)
    assert t.data_size() == ((16 + bw - 1)//bw)*((16 + bh - 1)//bh)*bs
    assert t.format() == fmt and t.type() == typ


def test_black_image_values_of_a_few_uncompressed_formats():
    for fmt, typ, want in ((Format.R8G8B8A8, U, [0, 0, 0, 255]), (Format.A8B8G8R8, U, [255, 0, 0, 0]),
                           (Format.R5G5B5A1, U, [1, 0]), (Format.A2B10G10R10, U, [0, 0, 0, 0xC0])):
        t = Texture(16, 16)
        assert t.set_image(black()) and t.convert(fmt, typ)
        assert np.array_equal(np.asarray(t.data()).reshape(256, -1), np.tile(np.array(want, np.uint8), (256, 1)))


def test_pairs_createConverter_rejects_return_false():
    legal = {(int(f), int(t)) for f, t in CASES}
    tried = 0
    for f in list(range(1, 29)) + [int(x) for x, _ in COMPRESSED]:
        for typ in Type:
            if (f, int(typ)) in legal:
                continue
            t = Texture(16, 16)
            assert t.set_image(black())
            assert not t.convert(Format(f), typ), (Format(f), typ)
            assert not t.converted()
            tried += 1
    assert tried > 150


def test_srgb_images_convert_only_to_formats_with_native_srgb():
    """TextureTest.cpp:815-820: a 15x10 sRGB texture refuses R5G6B5."""
    img = np.zeros((10, 15, 4), np.float32)
    for fmt, ok in ((Format.R5G6B5, False), (Format.R8G8B8A8, True), (Format.B8G8R8, True),
                    (Format.R16G16B16A16, False), (Format.BC7, True), (Format.BC4, False),
                    (Format.ETC1, False), (Format.ETC2_R8G8B8, True), (Format.ASTC_6x6, True)):
        t = Texture(15, 10, color_space=ColorSpace.sRGB)
        assert t.set_image(img)
        assert t.convert(fmt, U) == ok, fmt


def test_astc_hdr_profile_request_keeps_the_range():
    """Type::UFloat selects astcenc's HDR profiles (AstcConverter.cpp:150-162): values above 1 must
    come back (round 1 clamped them, round 2 first refused the pair)."""
    from cuttlefish_amd import synth
    img = synth.hdr_probe(36, 24, seed=4).astype(np.float32)*np.float32(4.0)   # up to 2.4e5: beyond half range
    img[:6, :6, :3] = -0.5                                                      # and below zero (one whole block)
    img[..., 3] = 1.0
    t = Texture(36, 24)
    assert t.set_image(img) and t.convert(Format.ASTC_6x6, UF)
    dec, bad = O.decode_astc_hdr(np.asarray(t.data()), int(Format.ASTC_6x6), 36, 24)
    assert bad == 0
    ref = np.maximum(img[..., :3], 0.0)
    got = dec[..., :3].astype(np.float32)
    assert got.max() == 65504.0 and got[:6, :6].max() == 0.0
    big = (ref > 1.0) & (ref < 6.0e4)
    assert np.median(np.abs(got[big] - ref[big])/ref[big]) < 0.12
    assert np.all(dec[..., 3].astype(np.float32) == 1.0)


def test_half_float_images_convert_to_every_block_format():
    """The reference turns every image into RGBAF first, so a float16 image is a legal source
    for every (format, type) pair -- not only for BC6H."""
    rng = np.random.default_rng(9)
    img16 = rng.random((16, 16, 4)).astype(np.float16)
    for fmt, typ in ((Format.BC7, U), (Format.BC1_RGB, U), (Format.ETC2_R8G8B8, U),
                     (Format.ASTC_6x6, U), (Format.BC6H, UF), (Format.R8G8B8A8, U)):
        a, b = Texture(16, 16), Texture(16, 16)
        assert a.set_image(img16) and b.set_image(img16.astype(np.float32))
        assert a.convert(fmt, typ) and b.convert(fmt, typ), fmt
        assert np.array_equal(np.asarray(a.data()), np.asarray(b.data())), fmt
