"""BC6H: decoder pinned to Pillow (all 14 modes, UF16 + SF16), encoder validity + PSNR."""
import ctypes
import os

import numpy as np
import pytest

import oracle_lib as O
from cuttlefish_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "pillow_bc6h.npz")
BC6H, UFLOAT, FLOAT = 35, 4, 5


def _to8(h):
    """Pillow's half -> 8-bit: (UINT8)(clamp(f,0,1)*255.0f) in float32."""
    f = h.view(np.float16).astype(np.float32)
    with np.errstate(invalid="ignore"):
        v = np.where(f < 0, 0, np.where(f > 1, 255,
                                        (np.clip(f, 0, 1) * np.float32(255.0)).astype(np.uint8)))
    return v.astype(np.uint8)


def _decode_blocks(blocks, flags):
    L = O.lib()
    out = np.zeros((blocks.shape[0], 48), np.uint16)
    for i in range(blocks.shape[0]):
        b = np.ascontiguousarray(blocks[i])
        L.cfo_decode_bc6h(ctypes.c_void_p(b.ctypes.data), flags, ctypes.c_void_p(out[i].ctypes.data))
    return out


@pytest.mark.parametrize("name,flags", [("uf16", 2), ("sf16", 3)])
def test_all_14_mode_layouts_match_pillow(name, flags):
    """flags bit 1 reproduces Pillow's two deviations from the D3D spec (documented in
    oracle/bc6h_decode.c); every bit field of every mode must then agree exactly."""
    d = np.load(GOLD)
    B, P = d[name + "_blocks"], d[name + "_pixels"]
    out = _decode_blocks(B, flags)
    e = (out >> 10) & 31
    nan = ((e == 31) & ((out & 1023) != 0)).reshape(-1, 16, 3).any(axis=2)
    got = _to8(out).reshape(-1, 16, 3)
    ok = (got == P).all(axis=2) | nan
    assert ok.all(), "modes with mismatches: %s" % sorted(set(np.flatnonzero(~ok.all(axis=1)) // 64 + 1))
    assert (~nan).mean() > 0.5


def test_float_to_half_is_round_to_nearest_even():
    """HalfFloatTest.cpp:34-68 known answer + a sweep against numpy's RNE conversion."""
    L = O.lib()
    L.cfo_float_to_half.restype = ctypes.c_uint16
    L.cfo_float_to_half.argtypes = [ctypes.c_float]
    ref = np.array([1.2, -3.4, 5.6, -7.8], np.float32)
    assert [L.cfo_float_to_half(float(v)) for v in ref] == list(ref.astype(np.float16).view(np.uint16))
    rng = np.random.default_rng(1)
    vals = (rng.standard_normal(4000) * 10.0 ** rng.integers(-9, 5, 4000)).astype(np.float32)
    vals = np.concatenate([vals, np.array([0, 65504, 65519.9, 65520, 5.96e-8, 2.98e-8, 6.1e-5],
                                          np.float32)])
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).view(np.uint16)
    got = np.array([L.cfo_float_to_half(float(v)) for v in vals], np.uint16)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("signed,typ", [(False, UFLOAT), (True, FLOAT)])
def test_encoder_quality_ladder(signed, typ):
    img = synth.hdr_probe(64, 64, seed=4, signed=signed)
    last = 0.0
    for q in (0, 1, 2, 4):
        blk = O.encode(img, BC6H, typ=typ, quality=q, threads=4)
        assert blk.nbytes == 16 * 16 * 16
        p = synth.psnr_log(img, O.decode_bc6h(blk, 64, 64, typ))
        assert p >= last - 0.05, (q, p, last)
        last = p
    assert last > 50.0


def test_reference_black_image_size_contract_and_solid_blocks():
    img = np.zeros((16, 16, 4), np.float32)
    img[..., 3] = 1.0
    blk = O.encode(img, BC6H, typ=UFLOAT)
    assert blk.nbytes == 4 * 4 * 16
    assert (O.decode_bc6h(blk, 16, 16, UFLOAT).view(np.uint16) == 0).all()
    for val in (0.5, 1.0, 1000.0, 60000.0):
        img[..., :3] = val
        dec = O.decode_bc6h(O.encode(img, BC6H, typ=UFLOAT), 16, 16, UFLOAT).astype(np.float64)
        assert np.abs(dec / val - 1.0).max() < 2e-3


def test_unsigned_clamps_negative_and_nonfinite():
    img = np.zeros((4, 4, 4), np.float32)
    img[..., 0] = -5.0
    img[..., 1] = np.inf
    img[..., 2] = 3.0
    dec = O.decode_bc6h(O.encode(img, BC6H, typ=UFLOAT), 4, 4, UFLOAT).astype(np.float64)
    assert (dec[..., 0] == 0).all()
    assert (dec[..., 1] > 65000).all() and np.isfinite(dec).all()
    assert np.abs(dec[..., 2] - 3.0).max() < 0.01


def test_half_source_is_passed_through_bit_exactly():
    """RGBA16F input == RGBA32F input holding the same values (RNE of exact halves)."""
    img = synth.hdr_probe(32, 20, seed=9)
    a = O.encode(img, BC6H, typ=UFLOAT, quality=2)
    b = O.encode(img.astype(np.float32), BC6H, typ=UFLOAT, quality=2)
    assert np.array_equal(a, b)
