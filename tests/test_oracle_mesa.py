"""Oracle decoders against an INDEPENDENT decoder: Mesa 23.2.1's software texture decompression
(fixture tests/golden/mesa_blocks.npz, generator make_mesa_fixtures.py).  This is what pins the
ETC2 / EAC bitstreams (Pillow has no ETC decoder), BC6H's half-float VALUES (Pillow only shows an
8-bit tone-clamped view) and, a second time, BC7.  The live tests at the bottom run the oracle's
ENCODERS through Mesa where the driver file exists (this image; skipped elsewhere)."""
import os

import numpy as np
import pytest

import mesa_lib as M
import oracle_lib as O
from cuttlefish_amd import synth

FIX = np.load(os.path.join(os.path.dirname(__file__), "golden", "mesa_blocks.npz"))


def _dims(blocks, bs):
    n = blocks.size // bs
    return 128, 4 * n // 32


@pytest.mark.parametrize("name,fmt,bs", [("etc2_rgb", 38, 8), ("etc2_rgba1", 39, 8),
                                         ("etc2_rgba8", 40, 16)])
def test_etc2_decoder_equals_mesa_on_random_blocks(name, fmt, bs):
    blk = FIX[name + "_blocks"]
    w, h = _dims(blk, bs)
    assert np.array_equal(O.decode_etc(blk, fmt, w, h), FIX[name + "_rgba"])


def test_etc2_fixture_covers_every_mode():
    """individual, differential, T, H, planar (overflow of R, G, B selects the last three)."""
    b = FIX["etc2_rgb_blocks"].reshape(-1, 8)
    diff = (b[:, 3] & 2) != 0
    def ovf(byte):
        base = (byte >> 3).astype(int)
        d = (byte & 7).astype(int)
        d = np.where(d >= 4, d - 8, d)
        return (base + d < 0) | (base + d > 31)
    t = diff & ovf(b[:, 0])
    hh = diff & ~ovf(b[:, 0]) & ovf(b[:, 1])
    pl = diff & ~ovf(b[:, 0]) & ~ovf(b[:, 1]) & ovf(b[:, 2])
    assert (~diff).sum() > 50 and t.sum() > 10 and hh.sum() > 10 and pl.sum() > 10
    assert (diff & ~t & ~hh & ~pl).sum() > 50


@pytest.mark.parametrize("name,fmt,bs", [("eac_r11", 41, 8), ("eac_rg11", 42, 16)])
@pytest.mark.parametrize("typ,tn", [(0, "u"), (1, "s")])
def test_eac11_decoder_equals_mesa(name, fmt, bs, typ, tn):
    blk = FIX["%s_%s_blocks" % (name, tn)]
    w, h = _dims(blk, bs)
    v = O.decode_eac(blk, fmt, w, h, typ).astype(np.int64)
    if typ == 0:      # 11 -> 16 bit replication of the specification (ETC2 spec, R11 EAC)
        e = (v << 5) | (v >> 6)
    else:             # signed: magnitude replicated, sign restored
        m = np.abs(v)
        e = np.sign(v) * ((m << 5) | (m >> 5))
    assert np.array_equal(e, FIX["%s_%s_px16" % (name, tn)].astype(np.int64))


def test_bc7_decoder_equals_mesa():
    blk = FIX["bc7_blocks"]
    w, h = _dims(blk, 16)
    assert np.array_equal(O.decode(blk, 36, w, h), FIX["bc7_rgba"])


@pytest.mark.parametrize("typ,tn", [(4, "uf16"), (5, "sf16")])
def test_bc6h_decoder_half_values_equal_mesa(typ, tn):
    blk = FIX["bc6h_%s_blocks" % tn]
    w, h = _dims(blk, 16)
    got = O.decode_bc6h(blk, w, h, typ).view(np.uint16)
    assert np.array_equal(got, FIX["bc6h_%s_half" % tn])


# ---- live: encoder output through Mesa --------------------------------------------------

live = pytest.mark.skipif(not M.available(), reason="Mesa swrast driver not in this environment")


@live
@pytest.mark.parametrize("fmt", [37, 38, 39, 40])
def test_etc_encoder_output_decodes_identically_under_mesa(fmt):
    img = synth.photo(64, 48, seed=40 + fmt)
    if fmt == 39:
        img[8:24, 8:40, 3] = 0
    blk = O.encode(img, fmt, quality=2, threads=4)
    assert np.array_equal(O.decode_etc(blk, fmt, 64, 48), M.decode(fmt, blk, 64, 48))


@live
@pytest.mark.parametrize("fmt", range(43, 57))
def test_astc_encoder_output_decodes_identically_under_mesa(fmt):
    img = synth.photo(72, 60, seed=fmt)
    for q in (0, 2, 3):
        blk = O.encode(img, fmt, quality=q, threads=4)
        mine, bad = O.decode_astc(blk, fmt, 72, 60)
        assert bad == 0
        assert np.array_equal(mine, M.decode(fmt, blk, 72, 60))


# ---- ASTC: the full LDR decoder against Mesa (fixture: make_mesa_astc_fixture.py) ----------

AFIX = np.load(os.path.join(os.path.dirname(__file__), "golden", "mesa_astc.npz"))
FOOTPRINTS = [(4, 4), (5, 4), (5, 5), (6, 5), (6, 6), (8, 5), (8, 6), (8, 8), (10, 5), (10, 6),
              (10, 8), (10, 10), (12, 10), (12, 12)]


@pytest.mark.parametrize("fi", range(14))
def test_astc_decoder_equals_mesa_on_structured_random_blocks(fi):
    """1-4 partitions (hash), dual plane, all LDR endpoint modes, trit/quint sequences, void
    extent, illegal encodings and HDR modes in the LDR profile (error colour, per partition)."""
    bw, bh = FOOTPRINTS[fi]
    blk = AFIX["blocks_%dx%d" % (bw, bh)]
    want = AFIX["rgba_%dx%d" % (bw, bh)]
    n = blk.shape[0]
    got, _ = O.decode_astc(blk.reshape(-1), 43 + fi, bw*n, bh)
    got = got.reshape(bh, n, bw, 4).transpose(1, 0, 2, 3).reshape(n, bh*bw, 4)
    assert np.array_equal(got, want)


def test_astc_fixture_covers_the_format_features():
    parts = np.zeros(5, int)
    dual = trit = quint = 0
    cems = set()
    for bw, bh in FOOTPRINTS:
        blk = AFIX["blocks_%dx%d" % (bw, bh)][:160]            # the valid ones
        lo = blk.view(np.uint64).reshape(-1, 2)[:, 0]
        for v in lo:
            v = int(v)
            if (v & 0x1FF) == 0x1FC:
                continue
            p = ((v >> 11) & 3) + 1
            parts[p] += 1
            dual += (v >> 10) & 1 if (v & 3) or ((v >> 7) & 3) != 2 else 0
            cems.add((v >> 13) & 15 if p == 1 else -1)
            r = ((v >> 4) & 1) | ((v & 3) << 1 if v & 3 else ((v >> 2) & 3) << 1)
            trit += r in (3, 6)
            quint += r == 5
    assert parts[1] > 300 and parts[2] > 200 and parts[3] > 100 and parts[4] > 50
    assert dual > 200 and trit > 200 and quint > 100
    assert {0, 1, 4, 5, 6, 8, 9, 10, 12, 13} <= cems
