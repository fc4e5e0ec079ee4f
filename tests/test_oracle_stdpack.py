"""oracle/std_pack.c against hand-derived known answers and independent numpy restatements of the
reference's uncompressed converters (lib/src/StandardConverter.h / .cpp, Converter.cpp:38-337).
The reference holds no value-level vectors for these (TextureTest.cpp:873-975 checks sizes)."""
import numpy as np
import pytest

import oracle_lib as O

UNORM, SNORM, UINT, INT, UFLOAT, FLOAT = range(6)
F = dict(R4G4=1, R4G4B4A4=2, B4G4R4A4=3, A4R4G4B4=4, R5G6B5=5, B5G6R5=6, R5G5B5A1=7, B5G5R5A1=8,
         A1R5G5B5=9, R8=10, R8G8=11, R8G8B8=12, B8G8R8=13, R8G8B8A8=14, B8G8R8A8=15, A8B8G8R8=16,
         A2R10G10B10=17, A2B10G10R10=18, R16=19, R16G16=20, R16G16B16=21, R16G16B16A16=22, R32=23,
         R32G32=24, R32G32B32=25, R32G32B32A32=26, B10G11R11=27, E5B9G9R9=28)

# createConverter's table, Converter.cpp:38-337: format -> {type: bytes per pixel}
LEGAL = {}
for n in ("R4G4",):
    LEGAL[F[n]] = {UNORM: 1}
for n in ("R4G4B4A4", "B4G4R4A4", "A4R4G4B4", "R5G6B5", "B5G6R5", "R5G5B5A1", "B5G5R5A1", "A1R5G5B5"):
    LEGAL[F[n]] = {UNORM: 2}
for i, n in enumerate(("R8", "R8G8", "R8G8B8", "R8G8B8A8")):
    LEGAL[F[n]] = {t: i + 1 for t in (UNORM, SNORM, UINT, INT)}
LEGAL[F["B8G8R8"]] = {UNORM: 3}
LEGAL[F["B8G8R8A8"]] = {UNORM: 4}
LEGAL[F["A8B8G8R8"]] = {UNORM: 4}
LEGAL[F["A2R10G10B10"]] = {UNORM: 4, UINT: 4}
LEGAL[F["A2B10G10R10"]] = {UNORM: 4, UINT: 4}
for i, n in enumerate(("R16", "R16G16", "R16G16B16", "R16G16B16A16")):
    LEGAL[F[n]] = {t: 2*(i + 1) for t in (UNORM, SNORM, UINT, INT, FLOAT)}
for i, n in enumerate(("R32", "R32G32", "R32G32B32", "R32G32B32A32")):
    LEGAL[F[n]] = {t: 4*(i + 1) for t in (UINT, INT, FLOAT)}
LEGAL[F["B10G11R11"]] = {UFLOAT: 4}
LEGAL[F["E5B9G9R9"]] = {UFLOAT: 4}

ALL_PAIRS = [(f, t) for f in sorted(LEGAL) for t in sorted(LEGAL[f])]


def test_legality_and_pixel_sizes_follow_createConverter():
    for f in range(0, 30):
        for t in range(6):
            assert O.std_pixel_bytes(f, t) == LEGAL.get(f, {}).get(t, 0), (f, t)


def px(*rgba):
    return np.array([[list(rgba)]], np.float32)


def word(b):
    return int.from_bytes(bytes(b), "little")


def test_known_answers_of_the_bit_field_packers():
    # r = 1 -> all ones, g = 0.5 -> round(0.5*max) (half away from zero), b = 0, a = 1
    p = px(1.0, 0.5, 0.0, 1.0)
    assert word(O.std_pack(p, F["R4G4"], UNORM)) == (8 | (15 << 4))                  # g | r << 4
    assert word(O.std_pack(p, F["R4G4B4A4"], UNORM)) == (15 | (0 << 4) | (8 << 8) | (15 << 12))
    assert word(O.std_pack(p, F["B4G4R4A4"], UNORM)) == (15 | (15 << 4) | (8 << 8) | (0 << 12))
    assert word(O.std_pack(p, F["A4R4G4B4"], UNORM)) == (0 | (8 << 4) | (15 << 8) | (15 << 12))
    assert word(O.std_pack(p, F["R5G6B5"], UNORM)) == (0 | (32 << 5) | (31 << 11))     # 31.5 -> 32
    assert word(O.std_pack(p, F["B5G6R5"], UNORM)) == (31 | (32 << 5) | (0 << 11))
    assert word(O.std_pack(p, F["R5G5B5A1"], UNORM)) == (1 | (0 << 1) | (16 << 6) | (31 << 11))  # 15.5 -> 16
    assert word(O.std_pack(p, F["B5G5R5A1"], UNORM)) == (1 | (31 << 1) | (16 << 6) | (0 << 11))
    assert word(O.std_pack(p, F["A1R5G5B5"], UNORM)) == (0 | (16 << 5) | (31 << 10) | (1 << 15))
    assert list(O.std_pack(p, F["B8G8R8"], UNORM)) == [0, 128, 255]                   # 127.5 -> 128
    assert list(O.std_pack(p, F["B8G8R8A8"], UNORM)) == [0, 128, 255, 255]
    assert list(O.std_pack(p, F["A8B8G8R8"], UNORM)) == [255, 0, 128, 255]
    assert word(O.std_pack(p, F["A2R10G10B10"], UNORM)) == (0 | (512 << 10) | (1023 << 20) | (3 << 30))
    assert word(O.std_pack(p, F["A2B10G10R10"], UNORM)) == (1023 | (512 << 10) | (0 << 20) | (3 << 30))
    q = px(1000.4, 7.5, 2000.0, 2.5)                                                   # UInt: clamp, round
    assert word(O.std_pack(q, F["A2R10G10B10"], UINT)) == (1023 | (8 << 10) | (1000 << 20) | (3 << 30))
    assert word(O.std_pack(q, F["A2B10G10R10"], UINT)) == (1000 | (8 << 10) | (1023 << 20) | (3 << 30))


def test_known_answers_of_the_channel_arrays():
    p = px(-1.0, 0.5, 2.0, -0.25)
    assert list(O.std_pack(p, F["R8G8B8A8"], UNORM)) == [0, 128, 255, 0]
    assert list(O.std_pack(p, F["R8G8B8A8"], SNORM).view(np.int8)) == [-127, 64, 127, -32]   # 63.5 -> 64, -31.75 -> -32
    assert list(O.std_pack(p, F["R16G16B16A16"], UNORM).view(np.uint16)) == [0, 32768, 65535, 0]
    assert list(O.std_pack(p, F["R16G16B16A16"], SNORM).view(np.int16)) == [-32767, 16384, 32767, -8192]
    q = px(-3.5, 2.5, 300.0, 70000.0)
    assert list(O.std_pack(q, F["R8G8B8A8"], UINT)) == [0, 3, 255, 255]
    assert list(O.std_pack(q, F["R8G8B8A8"], INT).view(np.int8)) == [-4, 3, 127, 127]      # ties away from zero
    assert list(O.std_pack(q, F["R16G16B16A16"], UINT).view(np.uint16)) == [0, 3, 300, 65535]
    assert list(O.std_pack(q, F["R16G16B16A16"], INT).view(np.int16)) == [-4, 3, 300, 32767]
    assert list(O.std_pack(q, F["R32G32B32A32"], UINT).view(np.uint32)) == [0, 3, 300, 70000]
    assert list(O.std_pack(q, F["R32G32B32A32"], INT).view(np.int32)) == [-4, 3, 300, 70000]
    assert np.array_equal(O.std_pack(q, F["R32G32B32A32"], FLOAT).view(np.float32), q.ravel())
    # HalfFloatTest.cpp:34-68's vector, through the R16G16B16A16 Float converter
    h = px(1.2, -3.4, 5.6, -7.8)
    assert np.array_equal(O.std_pack(h, F["R16G16B16A16"], FLOAT).view(np.float16),
                          h.ravel().astype(np.float16))
    # defined corners: NaN -> 0, casts saturate
    n = px(np.nan, np.inf, -np.inf, 5e9)
    assert list(O.std_pack(n, F["R32G32B32A32"], UINT).view(np.uint32)) == [0, 0xFFFFFFFF, 0, 0xFFFFFFFF]
    assert list(O.std_pack(n, F["R32G32B32A32"], INT).view(np.int32)) == [0, 0x7FFFFFFF, -0x80000000, 0x7FFFFFFF]
    assert list(O.std_pack(n, F["R8G8B8A8"], UNORM)) == [0, 255, 0, 255]


def round_half_away(x):
    x = np.asarray(x, np.float32)
    t = np.trunc(x)
    return t + np.where(np.abs(x - t) >= np.float32(0.5), np.copysign(np.float32(1), x), np.float32(0))


def test_channel_arrays_match_a_numpy_restatement_on_random_values():
    rng = np.random.default_rng(7)
    img = (rng.random((33, 45, 4)).astype(np.float32)*2.6 - 1.3)
    img[0, :8, 0] = np.arange(8, dtype=np.float32)/np.float32(510.0) + np.float32(0.5/255.0)
    for bits, dt_u, dt_s, fmts in ((8, np.uint8, np.int8, ("R8", "R8G8", "R8G8B8", "R8G8B8A8")),
                                   (16, np.uint16, np.int16, ("R16", "R16G16", "R16G16B16", "R16G16B16A16"))):
        umax, smax = np.float32(2**bits - 1), np.float32(2**(bits - 1) - 1)
        for c, name in enumerate(fmts, 1):
            sub = img[..., :c]
            want = round_half_away(np.clip(sub, 0, 1)*umax).astype(dt_u)
            assert np.array_equal(O.std_pack(img, F[name], UNORM).view(dt_u), want.ravel())
            want = round_half_away(np.clip(sub, -1, 1)*smax).astype(dt_s)
            assert np.array_equal(O.std_pack(img, F[name], SNORM).view(dt_s), want.ravel())
            big = sub*np.float32(40000.0)
            bigimg = img*np.float32(40000.0)
            want = round_half_away(np.clip(big, 0, umax)).astype(dt_u)
            assert np.array_equal(O.std_pack(bigimg, F[name], UINT).view(dt_u), want.ravel())
            want = round_half_away(np.clip(big, -smax - 1, smax)).astype(dt_s)
            assert np.array_equal(O.std_pack(bigimg, F[name], INT).view(dt_s), want.ravel())
            if bits == 16:
                assert np.array_equal(O.std_pack(img, F[name], FLOAT).view(np.float16),
                                      sub.astype(np.float16).ravel())


def test_half_converter_is_round_to_nearest_even_over_all_exponents():
    rng = np.random.default_rng(3)
    bits = rng.integers(0, 2**32, size=(64, 64, 4), dtype=np.uint32)
    f = bits.view(np.float32)
    f = np.where(np.isnan(f), np.float32(1.0), f)
    got = O.std_pack(f, F["R16G16B16A16"], FLOAT).view(np.uint16)
    with np.errstate(over="ignore"):
        assert np.array_equal(got, f.astype(np.float16).view(np.uint16).ravel())


def decode_11(v, mbits):
    e, m = v >> mbits, v & ((1 << mbits) - 1)
    return np.where(e == 0, m*2.0**(-14 - mbits), (1 + m/2.0**mbits)*2.0**(e.astype(np.float64) - 15))


def test_b10g11r11_truncates_toward_zero_in_the_normal_range():
    rng = np.random.default_rng(11)
    img = np.exp2(rng.uniform(-13.5, 15.5, size=(40, 40, 4))).astype(np.float32)
    w = O.std_pack(img, F["B10G11R11"], UFLOAT).view(np.uint32).reshape(40, 40)
    for c, (shift, ebits_m) in enumerate(((0, 6), (11, 6), (22, 5))):
        v = (w >> shift) & ((1 << (ebits_m + 5)) - 1)
        d = decode_11(v, ebits_m)
        x = img[..., c].astype(np.float64)
        assert np.all(d <= x) and np.all(x - d < x*2.0**(-ebits_m))
    z = px(0.0, np.inf, 1.0, 0.0)
    assert word(O.std_pack(z, F["B10G11R11"], UFLOAT)) == (0 | ((31 << 6) << 11) | ((15 << 5) << 22))


def test_e5b9g9r9_round_trips_within_half_a_step_and_clamps():
    rng = np.random.default_rng(13)
    img = np.exp2(rng.uniform(-18, 14.9, size=(48, 48, 4))).astype(np.float32)
    img[..., 1] *= rng.random((48, 48)).astype(np.float32)
    w = O.std_pack(img, F["E5B9G9R9"], UFLOAT).view(np.uint32).reshape(48, 48)
    e = (w >> 27).astype(np.float64)
    step = 2.0**(e - 24)
    for c in range(3):
        q = (w >> (9*c)) & 0x1FF
        x = np.minimum(img[..., c].astype(np.float64), 32768.0)
        assert np.all(np.abs(q*step - x) <= step*0.5000001)
    # the largest mantissa of the block uses the top bit unless the whole block is tiny
    m = np.maximum.reduce([(w >> (9*c)) & 0x1FF for c in range(3)])
    assert np.all((m >= 256) | (e == 0))
    assert word(O.std_pack(px(0, 0, 0, 1), F["E5B9G9R9"], UFLOAT)) == 0
    assert word(O.std_pack(px(-1.0, np.nan, 1.0, 1), F["E5B9G9R9"], UFLOAT)) == ((256 << 18) | (16 << 27))
    assert word(O.std_pack(px(1e9, 0, 0, 1), F["E5B9G9R9"], UFLOAT)) == (256 | (31 << 27))   # glm's SharedExpMax 2^15


def test_sources_other_than_float_see_the_reference_rgbaf_view():
    rng = np.random.default_rng(5)
    u8 = rng.integers(0, 256, size=(9, 13, 4), dtype=np.uint8)
    assert np.array_equal(O.std_pack(u8, F["R8G8B8A8"], UNORM), u8.ravel())                # exact round trip
    assert np.array_equal(O.std_pack(u8, F["B8G8R8A8"], UNORM).reshape(-1, 4), u8.reshape(-1, 4)[:, [2, 1, 0, 3]])
    h = (rng.random((9, 13, 4))*4 - 2).astype(np.float16)
    assert np.array_equal(O.std_pack(h, F["R16G16B16A16"], FLOAT).view(np.float16), h.ravel())


def test_negative_pitch_and_capacity():
    rng = np.random.default_rng(9)
    img = rng.random((7, 5, 4)).astype(np.float32)
    assert np.array_equal(O.std_pack(img[::-1], F["R5G6B5"], UNORM).view(np.uint16).reshape(7, 5),
                          O.std_pack(img, F["R5G6B5"], UNORM).view(np.uint16).reshape(7, 5)[::-1])
    with pytest.raises(ValueError):
        O.std_pack(img, F["R5G6B5"], FLOAT)


def test_the_kernels_division_free_unorm8_to_float_is_exact():
    # std_pack.hip: q = u*(1/255); q += fma(-q, 255, u)*(1/255) -- must equal u/255.0f for all u
    u = np.arange(256, dtype=np.float64)
    r = np.float64(np.float32(1.0)/np.float32(255.0))
    q = (u*r).astype(np.float32).astype(np.float64)               # float product (exact in double first)
    rem = (u - q*255.0).astype(np.float32).astype(np.float64)     # fma: one rounding of the exact value
    q2 = (rem*r + q).astype(np.float32)                           # fma (48-bit product + q: exact in double)
    assert np.array_equal(q2, np.arange(256, dtype=np.float32)/np.float32(255.0))
