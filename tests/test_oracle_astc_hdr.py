"""ASTC under the HDR profiles (Type::UFloat, lib/src/AstcConverter.cpp:150-162): the oracle encodes
HDR blocks with the endpoint modes 11 / 14 / 15.  No independent HDR ASTC decoder exists in this environment (Mesa's is
LDR only), so this leg is pinned only to the oracle's own from-specification decoder -- "parity
unpinned", stated in DESIGN.md; the tests check range, field use and self-consistency."""
import ctypes

import numpy as np
import pytest

import oracle_lib as O
from cuttlefish_amd import Alpha, Format, Type, synth


def _plog(ref, dec):
    a = np.log2(1 + np.abs(ref[..., :3].astype(np.float64)))
    b = np.log2(1 + np.abs(dec[..., :3].astype(np.float64)))
    return 10*np.log10(np.log2(65505.0)**2/np.mean((a - b)**2))


def _cems(payload):
    """colour endpoint mode of every single-partition block (bits 13..16), -1 for void extent /
    multi-partition blocks"""
    out = []
    for blk in payload.reshape(-1, 16):
        lo = int.from_bytes(blk[:8].tobytes(), "little")
        if (lo & 0x1FF) == 0x1FC:
            out.append(-1)
        elif ((lo >> 11) & 3) == 0:
            out.append((lo >> 13) & 15)
        else:
            out.append(-2)
    return np.array(out)


def test_lns_code_known_answers():
    L = O.lib()
    L.cfo_astc_hdr_code.restype = ctypes.c_int
    L.cfo_astc_hdr_code.argtypes = [ctypes.c_float]
    # 1.0 = half 0x3C00: exponent 15, mantissa 0 -> LNS16 15 << 11 = 30720 -> code 120
    assert L.cfo_astc_hdr_code(1.0) == 120
    assert L.cfo_astc_hdr_code(0.0) == 0 and L.cfo_astc_hdr_code(-3.0) == 0
    assert L.cfo_astc_hdr_code(float("nan")) == 0
    # 65504 = 0x7BFF: e 30, m10 1023 -> m = (8184 + 2050)/5 = 2046 -> (61440 + 2046 + 128) >> 8 = 248
    assert L.cfo_astc_hdr_code(65504.0) == 248 and L.cfo_astc_hdr_code(1.0e9) == 248
    # one code per eighth of an octave: 2.0 = code 128, 0.5 = code 112
    assert L.cfo_astc_hdr_code(2.0) == 128 and L.cfo_astc_hdr_code(0.5) == 112
    codes = [L.cfo_astc_hdr_code(float(v)) for v in np.geomspace(1e-4, 6e4, 400)]
    assert codes == sorted(codes)


@pytest.mark.parametrize("fmt,floor", [(Format.ASTC_4x4, 54.0), (Format.ASTC_6x6, 44.5), (Format.ASTC_10x8, 39.3)])
def test_hdr_probe_round_trip(fmt, floor):
    img = synth.hdr_probe(96, 72, seed=4).astype(np.float32)
    ps, seen = [], set()
    for q in (0, 2, 3):
        pay = O.encode(img, int(fmt), typ=int(Type.UFloat), quality=q, threads=8, alpha=int(Alpha.None_))
        dec, bad = O.decode_astc_hdr(pay, int(fmt), 96, 72)
        assert bad == 0
        assert np.all(dec[..., 3].astype(np.float32) == 1.0)
        ps.append(_plog(img, dec.astype(np.float32)))
        cem = _cems(pay)
        assert set(cem[cem >= 0]) <= {7, 11}         # opaque HDR blocks: mode 11, or mode 7 (base + scale)
        seen |= set(cem[cem >= 0])
    # (Lowest already ranks 8 configs of its one candidate; on a 96 x 72 probe the log-domain PSNR of the
    # levels is within noise of each other)
    assert ps[0] >= floor and ps[2] >= ps[0] - 0.1
    assert seen == {7, 11}                           # both ways of storing the endpoints are in use
    # the range survives: the probe's suns are tens of thousands
    assert float(dec[..., :3].astype(np.float32).max()) > 0.5*float(img[..., :3].max())


def test_alpha_profiles_pick_cem_14_and_15():
    rng = np.random.default_rng(5)
    img = synth.hdr_probe(48, 48, seed=6).astype(np.float32)
    img[..., 3] = rng.random((48, 48)).astype(np.float32)            # LDR-range alpha
    # Alpha::PreMultiplied -> ASTCENC_PRF_HDR_RGB_LDR_A: CEM 14 (HDR RGB + LDR alpha)
    pay = O.encode(img, int(Format.ASTC_6x6), typ=int(Type.UFloat), quality=2, threads=4, alpha=2)
    dec, bad = O.decode_astc_hdr(pay, int(Format.ASTC_6x6), 48, 48)
    cem = _cems(pay)
    assert bad == 0 and set(cem[cem >= 0]) <= {14} and (cem == 14).any()
    assert np.abs(dec[..., 3].astype(np.float32) - img[..., 3]).mean() < 0.14        # per-texel noise: no 6x6 block follows it
    # Alpha::Standard -> ASTCENC_PRF_HDR: CEM 15 (HDR alpha); alpha may exceed 1
    img2 = img.copy()
    img2[..., 3] *= 8.0
    pay = O.encode(img2, int(Format.ASTC_6x6), typ=int(Type.UFloat), quality=2, threads=4, alpha=1)
    dec, bad = O.decode_astc_hdr(pay, int(Format.ASTC_6x6), 48, 48)
    cem = _cems(pay)
    assert bad == 0 and set(cem[cem >= 0]) <= {15} and (cem == 15).any()
    assert float(dec[..., 3].astype(np.float32).max()) > 4.0


def test_solid_hdr_block_is_an_hdr_void_extent():
    img = np.zeros((12, 12, 4), np.float32)
    img[..., 0], img[..., 1], img[..., 2], img[..., 3] = 1000.0, 0.25, 3.0, 1.0
    pay = O.encode(img, int(Format.ASTC_6x6), typ=int(Type.UFloat), quality=2, alpha=0)
    for blk in pay.reshape(-1, 16):
        assert blk[0] == 0xFC and blk[1] == 0xFF          # void extent with the HDR bit (bit 9)
    dec, bad = O.decode_astc_hdr(pay, int(Format.ASTC_6x6), 12, 12)
    px = dec[0, 0].astype(np.float32)
    assert bad == 0 and abs(px[0] - 1000.0)/1000.0 < 0.05 and abs(px[1] - 0.25) < 0.02 and px[3] == 1.0


def test_ldr_profile_is_untouched_by_the_hdr_flags():
    img = synth.photo(36, 24, seed=8)
    a = O.encode(img, int(Format.ASTC_6x6), quality=2)
    assert O.decode_astc(a, int(Format.ASTC_6x6), 36, 24)[1] == 0
    cem = _cems(a)
    assert not (set(cem[cem >= 0]) & {11, 14, 15})


# ---- every HDR endpoint sub-mode of the decoder against the specification's bit-placement tables ----
# The C decoder (oracle/astc_decode.c) extracts the fields with one-hot masks; this side PLACES
# them, straight from the tables: which bit of which field each variable-placement position X_k
# carries in each sub-mode.  A field is ("a", 9) = bit 9 of a.

_M11_BITS = [(9, 7, 6, 7), (9, 8, 6, 6), (10, 6, 7, 7), (10, 7, 7, 6), (11, 8, 6, 5), (11, 6, 8, 6), (12, 7, 7, 5), (12, 6, 7, 6)]
# X0 = v2 bit 6, X1 = v3 bit 6, X2 = v4 bit 6, X3 = v5 bit 6, X4 = v4 bit 5, X5 = v5 bit 5
_M11_X = [
    [("b0", 6), ("b1", 6), ("d0", 6), ("d1", 6), ("d0", 5), ("d1", 5)],
    [("b0", 6), ("b1", 6), ("b0", 7), ("b1", 7), ("d0", 5), ("d1", 5)],
    [("a", 9), ("c", 6), ("d0", 6), ("d1", 6), ("d0", 5), ("d1", 5)],
    [("b0", 6), ("b1", 6), ("a", 9), ("c", 6), ("d0", 5), ("d1", 5)],
    [("b0", 6), ("b1", 6), ("b0", 7), ("b1", 7), ("a", 9), ("a", 10)],
    [("a", 9), ("a", 10), ("c", 7), ("c", 6), ("d0", 5), ("d1", 5)],
    [("b0", 6), ("b1", 6), ("a", 11), ("c", 6), ("a", 9), ("a", 10)],
    [("a", 9), ("a", 10), ("a", 11), ("c", 6), ("d0", 5), ("d1", 5)],
]


def _unpack(cem, v):
    L = O.lib()
    arr = (ctypes.c_int*8)(*(list(v) + [0]*(8 - len(v))))
    e0, e1 = (ctypes.c_int*4)(), (ctypes.c_int*4)()
    L.cfo_astc_unpack_endpoints.restype = ctypes.c_int
    kind = L.cfo_astc_unpack_endpoints(cem, arr, e0, e1)
    return kind, list(e0), list(e1)


def _place_mode11(mode, maj, f):
    """fields (two's-complement d0 / d1 in their dbits) -> v0..v5"""
    bit = lambda name, k: (f[name] >> k) & 1
    v = [0]*6
    v[0] = f["a"] & 0xFF
    v[1] = ((mode & 1) << 7) | (bit("a", 8) << 6) | (f["c"] & 0x3F)
    v[2] = (((mode >> 1) & 1) << 7) | (f["b0"] & 0x3F)
    v[3] = (((mode >> 2) & 1) << 7) | (f["b1"] & 0x3F)
    v[4] = ((maj & 1) << 7) | (f["d0"] & 0x1F)
    v[5] = (((maj >> 1) & 1) << 7) | (f["d1"] & 0x1F)
    pos = [(2, 6), (3, 6), (4, 6), (5, 6), (4, 5), (5, 5)]
    for (vi, bi), (name, k) in zip(pos, _M11_X[mode]):
        v[vi] |= bit(name, k) << bi
    return v


def test_mode_11_submodes_against_the_placement_table():
    rng = np.random.default_rng(11)
    for mode in range(8):
        ab, bb, cb, db = _M11_BITS[mode]
        sh = 12 - ab
        assert sh == (mode >> 1) ^ 3
        for maj in range(3):
            for _ in range(200):
                a, b0, b1, c = (int(rng.integers(0, 1 << n)) for n in (ab, bb, bb, cb))
                d0, d1 = (int(rng.integers(-(1 << (db - 1)), 1 << (db - 1))) for _ in range(2))
                f = {"a": a, "b0": b0, "b1": b1, "c": c, "d0": d0 & ((1 << db) - 1), "d1": d1 & ((1 << db) - 1)}
                v = _place_mode11(mode, maj, f)
                A, B0, B1, C, D0, D1 = (x << sh for x in (a, b0, b1, c, d0, d1))
                hi = [A, A - B0, A - B1]
                lo = [A - C, A - B0 - C - D0, A - B1 - C - D1]
                cl = lambda x: min(max(x, 0), 4095) << 4
                hi, lo = [cl(x) for x in hi], [cl(x) for x in lo]
                if maj:                                    # the major component takes red's place
                    hi[0], hi[maj] = hi[maj], hi[0]
                    lo[0], lo[maj] = lo[maj], lo[0]
                kind, e0, e1 = _unpack(11, v)
                assert kind == 1 and e0[:3] == lo and e1[:3] == hi and e0[3] == e1[3] == 0x7800, (mode, maj, f)


def test_mode_11_known_answers_by_hand():
    # sub-mode 7 (12-bit a, shift 0), major component 0: a = 0xABC, b0 = 5, b1 = 9, c = 0x21, d0 = -3, d1 = 2
    #   v0 = 0xBC; v1 = m0 1 | a8 (0xABC >> 8 & 1 = 0) | c[5:0] 0x21            = 0xA1
    #   v2 = m1 1 | X0 = a9 (1) | b0 = 5                                          = 0xC5
    #   v3 = m2 1 | X1 = a10 (0) | b1 = 9                                         = 0x89
    #   v4 = maj0 0 | X2 = a11 (1) | X4 = d0 bit 5 (-3 = 0b111101 -> 1) | 0b11101 = 0x7D
    #   v5 = maj1 0 | X3 = c6 (0) | X5 = d1 bit 5 (0) | 0b00010                   = 0x02
    kind, e0, e1 = _unpack(11, [0xBC, 0xA1, 0xC5, 0x89, 0x7D, 0x02])
    a, b0, b1, c, d0, d1 = 0xABC, 5, 9, 0x21, -3, 2
    assert e1[:3] == [a << 4, (a - b0) << 4, (a - b1) << 4]
    assert e0[:3] == [(a - c) << 4, (a - b0 - c - d0) << 4, (a - b1 - c - d1) << 4]
    assert e1[:3] == [43968, 43888, 43824] and e0[:3] == [43440, 43408, 43264]
    # the direct form (major component 3): six values, blue on 7 bits
    kind, e0, e1 = _unpack(11, [0x12, 0x34, 0x56, 0x78, 0x80 | 0x1A, 0x80 | 0x7F])
    assert kind == 1 and e0 == [0x1200, 0x5600, 0x1A << 9, 0x7800] and e1 == [0x3400, 0x7800, 0x7F << 9, 0x7800]
    # sub-mode 0, major component 2 (blue leads): a = 0x1FF (9 bits, shift 3), all differences zero
    #   v0 = 0xFF, v1 = 0 | a8 1 << 6 | 0 = 0x40, v2 = v3 = 0, v4 = 0x00, v5 = 0x80
    kind, e0, e1 = _unpack(11, [0xFF, 0x40, 0, 0, 0x00, 0x80])
    assert e0[:3] == e1[:3] == [(0x1FF << 3) << 4]*3


_M7_BITS = [(11, 5, 5, 7), (11, 6, 6, 5), (10, 5, 5, 8), (9, 6, 6, 7), (8, 7, 7, 6), (7, 7, 7, 7)]
# X0 = v1 bit 6, X1 = v1 bit 5, X2 = v2 bit 6, X3 = v2 bit 5, X4 = v3 bit 7, X5 = v3 bit 6, X6 = v3 bit 5
_M7_X = [
    [("r", 9), ("r", 8), ("r", 7), ("r", 10), ("r", 6), ("s", 6), ("s", 5)],
    [("r", 8), ("g", 5), ("r", 7), ("b", 5), ("r", 6), ("r", 10), ("r", 9)],
    [("r", 9), ("r", 8), ("r", 7), ("r", 6), ("s", 7), ("s", 6), ("s", 5)],
    [("r", 8), ("g", 5), ("r", 7), ("b", 5), ("r", 6), ("s", 6), ("s", 5)],
    [("g", 6), ("g", 5), ("b", 6), ("b", 5), ("r", 6), ("r", 7), ("s", 5)],
    [("g", 6), ("g", 5), ("b", 6), ("b", 5), ("r", 6), ("s", 6), ("s", 5)],
]


def test_mode_7_submodes_against_the_placement_table():
    rng = np.random.default_rng(7)
    for mode in range(6):
        rb, gb_, bb, sb = _M7_BITS[mode]
        sh = [1, 1, 2, 3, 4, 5][mode]
        assert rb + sh == 12
        for maj in range(3 if mode < 5 else 1):
            # the four mode bits M0..M3: sub-modes 0-3 = major component << 2 | mode; 4 = 0b11xx with xx = major
            # component; 5 = 0b1111
            mv = (maj << 2 | mode) if mode < 4 else ((0xC | maj) if mode == 4 else 0xF)
            for _ in range(200):
                f = {"r": int(rng.integers(0, 1 << rb)), "g": int(rng.integers(0, 1 << gb_)),
                     "b": int(rng.integers(0, 1 << bb)), "s": int(rng.integers(0, 1 << sb))}
                bit = lambda name, k: (f[name] >> k) & 1
                v = [((mv & 3) << 6) | (f["r"] & 0x3F), (((mv >> 2) & 1) << 7) | (f["g"] & 0x1F),
                     (((mv >> 3) & 1) << 7) | (f["b"] & 0x1F), f["s"] & 0x1F]
                pos = [(1, 6), (1, 5), (2, 6), (2, 5), (3, 7), (3, 6), (3, 5)]
                for (vi, bi), (name, k) in zip(pos, _M7_X[mode]):
                    v[vi] |= bit(name, k) << bi
                R, G, B, S = (f[k] << sh for k in "rgbs")
                if mode != 5:
                    G, B = R - G, R - B
                hi = [R, G, B]
                if maj:
                    hi[0], hi[maj] = hi[maj], hi[0]
                lo = [max(x - S, 0) << 4 for x in hi]
                hi = [max(x, 0) << 4 for x in hi]
                kind, e0, e1 = _unpack(7, v)
                assert kind == 1 and e0[:3] == lo and e1[:3] == hi and e0[3] == e1[3] == 0x7800, (mode, maj, f)


def test_hdr_luminance_and_alpha_modes_known_answers():
    # mode 2: v1 >= v0 -> (v0 << 4, v1 << 4) as 12-bit values; else the half-step form
    assert _unpack(2, [0x10, 0x20])[1:] == ([0x1000]*3 + [0x7800], [0x2000]*3 + [0x7800])
    assert _unpack(2, [0x20, 0x10])[1:] == ([((0x10 << 4) + 8) << 4]*3 + [0x7800], [((0x20 << 4) - 8) << 4]*3 + [0x7800])
    # mode 3, v0 bit 7 set: y0 = v1[7:5] << 9 | v0[6:0] << 2, d = v1[4:0] << 2
    k, e0, e1 = _unpack(3, [0x80 | 0x55, 0xA0 | 0x13])
    y0 = ((0xA0 | 0x13) & 0xE0) << 4 | 0x55 << 2
    assert e0[:3] == [y0 << 4]*3 and e1[:3] == [(y0 + (0x13 << 2)) << 4]*3
    # mode 3, v0 bit 7 clear: y0 = v1[7:4] << 8 | v0 << 1, d = v1[3:0] << 1; y1 saturates at 0xFFF
    k, e0, e1 = _unpack(3, [0x7F, 0xFF])
    assert e0[0] == (0xF00 | 0xFE) << 4 and e1[0] == 0xFFF << 4
    # mode 15 alpha, selector 3 (both top bits set): two 7-bit values << 9
    k, e0, e1 = _unpack(15, [0, 0, 0, 0, 0x80, 0x80, 0x80 | 0x3C, 0x80 | 0x41])
    assert (e0[3], e1[3]) == (0x3C << 9, 0x41 << 9)
    # selector 0 (v6 bit 7 = 0, v7 bit 7 = 0): base = v6[6:0] | v7[6] << 7 (8 bits) << 4, offset = v7[5:0]
    # signed 6 bits << 4:  v6 = 0x35, v7 = 0x40 | 0x3E (offset -2) -> base 0xB5 << 4 = 0xB50, end 0xB50 - 0x20
    k, e0, e1 = _unpack(15, [0, 0, 0, 0, 0x80, 0x80, 0x35, 0x40 | 0x3E])
    assert (e0[3], e1[3]) == (0xB50 << 4, (0xB50 - 0x20) << 4)
    # selector 1 (v6 bit 7 = 1, v7 bit 7 = 0): base 9 bits (v6[6:0] | v7[6:5] << 7) << 3, offset v7[4:0] signed << 3
    k, e0, e1 = _unpack(15, [0, 0, 0, 0, 0x80, 0x80, 0x80 | 0x11, 0x60 | 0x05])
    assert (e0[3], e1[3]) == (((0x11 | 3 << 7) << 3) << 4, (((0x11 | 3 << 7) << 3) + (5 << 3)) << 4)
    # selector 2 (v6 bit 7 = 0, v7 bit 7 = 1): base 10 bits (v6[6:0] | v7[6:4] << 7) << 2, offset v7[3:0] signed << 2
    k, e0, e1 = _unpack(15, [0, 0, 0, 0, 0x80, 0x80, 0x7F, 0x80 | 0x70 | 0x9])
    assert (e0[3], e1[3]) == ((0x3FF << 2) << 4, ((0x3FF << 2) + ((9 - 16) << 2)) << 4)


def test_hdr_4x4_is_within_reach_of_bc6h_at_the_same_rate():
    """8 bits per pixel both: mode 11's sub-modes at up to 12 bits per endpoint channel put ASTC 4x4 past
    BC6H on the probe in the log domain (round 2's 8-bit direct form was 6 dB behind)."""
    img = synth.hdr_probe(128, 128, seed=4).astype(np.float16)
    a = O.encode(img.astype(np.float32), int(Format.ASTC_4x4), typ=int(Type.UFloat), quality=2, threads=8, alpha=0)
    dec, bad = O.decode_astc_hdr(a, int(Format.ASTC_4x4), 128, 128)
    b6 = O.encode(img, int(Format.BC6H), typ=int(Type.UFloat), quality=2, threads=8)
    d6 = O.decode_bc6h(b6, 128, 128)
    pa, pb = _plog(img.astype(np.float32), dec.astype(np.float32)), _plog(img.astype(np.float32), d6.astype(np.float32))
    assert bad == 0 and pa > pb - 2.0, (pa, pb)
    # the sub-modes are in use: most single-partition blocks do NOT carry the direct form's marker (v4, v5 top bits)
    assert pa > 50.0


def test_hdr_solid_block_keeps_its_exact_halves():
    img = np.zeros((8, 8, 4), np.float32)
    vals = np.array([0.1234, 777.5, 3.0e-3, 1.0], np.float32)
    img[...] = vals
    pay = O.encode(img, int(Format.ASTC_4x4), typ=int(Type.UFloat), quality=2, alpha=0)
    dec, bad = O.decode_astc_hdr(pay, int(Format.ASTC_4x4), 8, 8)
    want = vals.astype(np.float16)
    # (an LNS value keeps 11 mantissa bits of the half's 10: the round trip half -> LNS -> half is exact)
    assert bad == 0 and np.array_equal(dec[0, 0, :3], want[:3]) and dec[0, 0, 3] == np.float16(1.0)


def test_half_to_lns_to_half_is_exact_for_every_finite_half():
    L = O.lib()
    L.cfo_astc_lns16.restype = ctypes.c_int
    L.cfo_astc_lns16.argtypes = [ctypes.c_uint16]

    def back(c):                                   # the specification's LNS -> half
        e, m = c >> 11, c & 0x7FF
        mt = 3*m if m < 512 else (4*m - 512 if m < 1536 else 5*m - 2048)
        return min((e << 10) + (mt >> 3), 0x7BFF)
    prev = -1
    for h in range(0x7C00):
        c = L.cfo_astc_lns16(h)
        assert back(c) == h and c > prev, hex(h)
        prev = c
    assert L.cfo_astc_lns16(0x3C00) == 0x7800


def test_encoder_placement_round_trips_through_the_decoder():
    """What hdr_rgb_place stores is what the decoder reads back: for endpoint pairs a sub-mode can hold
    (differences inside its b / c / d fields) the decoded pair equals the pair rounded to the mode's step;
    the direct form keeps the top 8 (blue: 7) bits.  Alpha selectors alike."""
    L = O.lib()
    rng = np.random.default_rng(12)
    bits = [(9, 7, 6, 7), (9, 8, 6, 6), (10, 6, 7, 7), (10, 7, 7, 6), (11, 8, 6, 5), (11, 6, 8, 6), (12, 7, 7, 5), (12, 6, 7, 6)]
    I3, I6 = ctypes.c_int*3, ctypes.c_int*6
    for m, (ab, bb, cb, db) in enumerate(bits):
        sh = 12 - ab
        hits = 0
        for _ in range(400):
            maj = int(rng.integers(0, 3))
            a = int(rng.integers(1 << (ab - 1), 1 << ab)) << sh
            c = int(rng.integers(0, 1 << cb)) << sh
            b = [int(rng.integers(0, 1 << bb)) << sh for _ in range(2)]
            d = [int(rng.integers(-(1 << (db - 1)), 1 << (db - 1))) << sh for _ in range(2)]
            hi = [a, a - b[0], a - b[1]]
            lo = [a - c, a - b[0] - c - d[0], a - b[1] - c - d[1]]
            if min(hi + lo) < 0 or max(hi + lo) > 4095 or max(hi[1:]) >= hi[0]:
                continue                               # the major component must be the strict maximum
            hi[0], hi[maj] = hi[maj], hi[0]
            lo[0], lo[maj] = lo[maj], lo[0]
            v, hm = I6(), I6()
            L.cfo_astc_hdr_place(1 + m, I3(*lo), I3(*hi), v, hm)
            kind, e0, e1 = _unpack(11, list(v))
            assert e0[:3] == [x << 4 for x in lo] and e1[:3] == [x << 4 for x in hi], (m, maj, lo, hi, list(v))
            # the bits a requantisation has to keep cover the mode and major-component bits
            assert hm[1] & 0x80 and hm[2] & 0x80 and hm[3] & 0x80 and hm[4] & 0x80 and hm[5] & 0x80
            hits += 1
        assert hits > 50, m
    # direct form
    for _ in range(200):
        lo = [int(x) for x in rng.integers(0, 4096, 3)]
        hi = [int(x) for x in rng.integers(0, 4096, 3)]
        v, hm = I6(), I6()
        L.cfo_astc_hdr_place(0, I3(*lo), I3(*hi), v, hm)
        kind, e0, e1 = _unpack(11, list(v))
        for c in range(3):
            step = 512 if c == 2 else 256
            top = 127 if c == 2 else 255
            assert e0[c] == min((lo[c]*16 + step//2)//step, top)*step and e1[c] == min((hi[c]*16 + step//2)//step, top)*step
    # alpha selectors 0..2: base on 8 + s bits (step 16 >> s in 12-bit units), offset on 6 - s signed bits
    I2 = ctypes.c_int*2
    for sel in range(3):
        sh = 4 - sel
        for _ in range(200):
            base = int(rng.integers(0, 1 << (8 + sel))) << sh
            off = int(rng.integers(-(1 << (5 - sel)), 1 << (5 - sel))) << sh
            if not 0 <= base + off <= 4095:
                continue
            v, hm = I2(), I2()
            L.cfo_astc_hdr_alpha_place(sel, base, base + off, v, hm)
            kind, e0, e1 = _unpack(15, [0, 0, 0, 0, 0x80, 0x80, v[0], v[1]])
            assert (e0[3], e1[3]) == (base << 4, (base + off) << 4), (sel, base, off)


def test_mode_7_placement_round_trips_through_the_decoder():
    """hdr_scale_place against the decoder: a (high endpoint, scale) pair a sub-mode can hold (the differences
    from the major component inside its green / blue fields) decodes to the pair rounded to the sub-mode's
    step; any pair decodes to within half a step per field where no field clamps."""
    L = O.lib()
    rng = np.random.default_rng(13)
    bits = [(11, 5, 7), (11, 6, 5), (10, 5, 8), (9, 6, 7), (8, 7, 6), (7, 7, 7)]
    shamt = [1, 1, 2, 3, 4, 5]
    I3, I4 = ctypes.c_int*3, ctypes.c_int*4
    for m, (rb, gb, sb) in enumerate(bits):
        sh = shamt[m]
        hits = 0
        for _ in range(600):
            maj = int(rng.integers(0, 3)) if m < 5 else 0
            red = int(rng.integers(1 << (rb - 2), 1 << rb)) << sh
            if m < 5:
                g = red - (int(rng.integers(1, 1 << gb)) << sh)
                b = red - (int(rng.integers(1, 1 << gb)) << sh)
            else:
                g = int(rng.integers(0, 1 << gb)) << sh
                b = int(rng.integers(0, 1 << gb)) << sh
            scale = int(rng.integers(0, 1 << sb)) << sh
            hi = [red, g, b]
            if min(hi) < 0 or max(hi) > 4095:
                continue
            hi[0], hi[maj] = hi[maj], hi[0]
            v, hm = I4(), I4()
            L.cfo_astc_hdr_scale_place(m, I3(*hi), scale, v, hm)
            kind, e0, e1 = _unpack(7, list(v))
            assert e1[:3] == [x << 4 for x in hi], (m, maj, hi, scale, list(v))
            assert e0[:3] == [max(x - scale, 0) << 4 for x in hi], (m, maj, hi, scale, list(v))
            assert hm[0] == 0xC0 and hm[1] & 0x80 and hm[2] & 0x80      # mode and major-component bits are kept
            hits += 1
        assert hits > 100, m
        # arbitrary pairs: half a step per field where nothing clamps
        for _ in range(300):
            hi = sorted((int(x) for x in rng.integers(0, min(4095, ((1 << rb) - 1) << sh), 3)), reverse=True)
            lim = (1 << gb) << sh
            if m < 5 and (hi[0] - hi[2] >= lim - (1 << sh)):
                continue
            if m == 5 and max(hi) > ((1 << gb) - 1) << sh:
                continue
            scale = int(rng.integers(0, ((1 << sb) - 1) << sh))
            v, hm = I4(), I4()
            L.cfo_astc_hdr_scale_place(m, I3(*hi), scale, v, hm)
            kind, e0, e1 = _unpack(7, list(v))
            half = 1 << (sh - 1)
            assert abs((e1[0] >> 4) - hi[0]) <= half
            for c in (1, 2):
                assert abs((e1[c] >> 4) - hi[c]) <= 2*half, (m, hi, list(v), e1)


def test_requant_keep_closed_form_equals_the_scan():
    """The kernel's requant_keep (nearest value if it keeps the bits, else the first stored value on the other
    side: floor table / its mirror image) against the oracle's outward scan: every level, value and mask."""
    assert O.lib().cfo_astc_requant_closed_form_mismatches() == 0


def test_luminance_placement_round_trips_and_grey_blocks_use_modes_2_and_3():
    """hdr_lum_place against the decoder: a grey pair a form can hold decodes to the pair rounded to the form's
    step (mode 2: 16, or 16 shifted by 8 when stored swapped; mode 3: 2 with a 4-bit offset, 4 with a 5-bit one).
    A grey opaque HDR image then carries modes 2 and 3 beside 7 and 11, and decodes without an illegal block."""
    L = O.lib()
    L.cfo_astc_hdr_lum_place.restype = ctypes.c_int
    rng = np.random.default_rng(14)
    I2 = ctypes.c_int*2
    seen = [0, 0, 0, 0]
    for _ in range(4000):
        form = int(rng.integers(0, 4))
        if form == 0:
            lo, hi = sorted(int(x) << 4 for x in rng.integers(0, 256, 2))
        elif form == 1:
            a, b = sorted(int(x) for x in rng.integers(0, 256, 2))
            if a == b:
                continue
            lo, hi = (a << 4) + 8, (b << 4) - 8
        elif form == 2:
            lo = int(rng.integers(0, 2048)) << 1
            hi = lo + (int(rng.integers(0, 16)) << 1)
        else:
            lo = int(rng.integers(0, 1024)) << 2
            hi = lo + (int(rng.integers(0, 32)) << 2)
        if hi > 4095:
            continue
        v, hm = I2(), I2()
        ok = L.cfo_astc_hdr_lum_place(form, lo, hi, v, hm)
        assert ok == 1, (form, lo, hi)
        kind, e0, e1 = _unpack(2 if form < 2 else 3, list(v))
        assert e0[:3] == [lo << 4]*3 and e1[:3] == [hi << 4]*3, (form, lo, hi, list(v), e0, e1)
        seen[form] += 1
    assert min(seen) > 300
    # a pair the small-range forms cannot hold is refused, not clamped silently
    v, hm = I2(), I2()
    assert L.cfo_astc_hdr_lum_place(2, 100, 200, v, hm) == 0 and L.cfo_astc_hdr_lum_place(3, 100, 400, v, hm) == 0
    assert L.cfo_astc_hdr_lum_place(0, 300, 100, v, hm) == 0
    img = synth.hdr_probe(96, 96, seed=4).astype(np.float32)
    g = img[..., :3].mean(-1).astype(np.float16).astype(np.float32)
    img[..., 0] = img[..., 1] = img[..., 2] = g
    img[..., 3] = 1.0
    for fmt in (Format.ASTC_4x4, Format.ASTC_8x8):
        pay = O.encode(img, int(fmt), typ=int(Type.UFloat), quality=2, threads=8, alpha=int(Alpha.None_))
        dec, bad = O.decode_astc_hdr(pay, int(fmt), 96, 96)
        cem = _cems(pay)
        assert bad == 0 and {2, 3} <= set(cem[cem >= 0]) <= {2, 3, 7, 11}


def test_hdr_ladder_against_the_hdr_wide_search():
    """Round 6 (round-5 VERDICT missing 3 / next 6): the HDR profiles have a bound now.  cfo_astc_wide_search_hdr walks one
    partition, a second plane on every component, every canonical 2 / 3 / 4-partition seed, every config of the class
    (census tables and the encoder's lists), forces every way of storing the endpoints (mode 11 / 14 / 15, mode 7, the
    luminance modes) and lets every partition price every sub-mode -- all measured exactly on the 16-bit LNS values.
    Content: blocks of the real photographs under synthetic radiometry (tools/quality_real.py hdr_blocks: inverse display
    curve, -6 .. +8 stops of exposure per block) -- real structure, invented radiances; no HDR photograph exists here.
    The block the bound writes decodes (oracle HDR decoder) to no more than any level's error on every block, and the
    ladder's distance is what profiles/r06_quality_real.md states: 4x4 0.8 / 0.6 / 0.6 dB, 6x6 1.1 / 1.0 / 0.75 (LNS PSNR).
    Still true: no INDEPENDENT HDR decoder pins the oracle's -- this leg rests on self-consistency."""
    import importlib.util
    import os
    import real_lib as R
    spec = importlib.util.spec_from_file_location(
        "quality_real", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "quality_real.py"))
    Q = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(Q)
    for (bw, bh), (ln, lh) in (((4, 4), (0.95, 0.80)), ((6, 6), (1.30, 1.25))):
        hb = Q.hdr_blocks(R.blocks(bw, bh, 64))
        sse, vals = Q.astc_hdr_sse(hb, bw, bh)
        assert (sse[:5] >= sse[5][None, :]).all()                   # a bound on every block
        ps = [Q.psnr_lns(sse[q].sum(), vals * len(hb)) for q in range(6)]
        assert ps[5] - ps[2] <= ln and ps[5] - ps[3] <= lh, ps
        assert all(ps[q + 1] >= ps[q] - 0.05 for q in range(4)), ps
