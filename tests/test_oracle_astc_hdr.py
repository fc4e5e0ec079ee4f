"""ASTC under the HDR profiles (Type::UFloat, lib/src/AstcConverter.cpp:150-162): the oracle encodes
HDR blocks with the endpoint modes 11 / 14 / 15 in their direct sub-mode, searching in the top 8
bits of the LNS domain.  No independent HDR ASTC decoder exists in this environment (Mesa's is
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


@pytest.mark.parametrize("fmt,floor", [(Format.ASTC_4x4, 46.0), (Format.ASTC_6x6, 40.0), (Format.ASTC_10x8, 33.0)])
def test_hdr_probe_round_trip(fmt, floor):
    img = synth.hdr_probe(96, 72, seed=4).astype(np.float32)
    ps = []
    for q in (0, 2, 3):
        pay = O.encode(img, int(fmt), typ=int(Type.UFloat), quality=q, threads=8, alpha=int(Alpha.None_))
        dec, bad = O.decode_astc_hdr(pay, int(fmt), 96, 72)
        assert bad == 0
        assert np.all(dec[..., 3].astype(np.float32) == 1.0)
        ps.append(_plog(img, dec.astype(np.float32)))
        cem = _cems(pay)
        assert set(cem[cem >= 0]) <= {11}            # opaque HDR blocks: HDR RGB direct
    # (Lowest already ranks 8 configs of its one candidate; on a 96 x 72 probe the log-domain PSNR of the
    # levels is within noise of each other -- the ladder is checked on the code-domain error the encoder minimises)
    assert ps[0] >= floor and ps[2] >= ps[0] - 0.25
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
    assert np.abs(dec[..., 3].astype(np.float32) - img[..., 3]).mean() < 0.12
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
