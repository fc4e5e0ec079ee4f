"""Mip-level resize oracle (oracle/mipgen.c), SURVEY section 8(f) row 1.

Pinning: the colour-space functions against the REFERENCE's own Color.h compiled into
oracle/_ref/libcf_ref.so (bit for bit, when that build is present); the resize loops against a
direct Python transcription of the fallback's index arithmetic and against closed forms.
"""
import math

import numpy as np
import pytest

import oracle_lib as O


def test_color_functions_equal_the_reference_build():
    R = O.ref_lib()
    if R is None:
        pytest.skip("oracle/_ref not built (needs /root/reference): make -C oracle ref")
    s2l, l2s = O.color_fns()
    rng = np.random.default_rng(5)
    xs = np.concatenate([np.linspace(0.0, 1.0, 4097), rng.random(20000), [0.04045, 0.0031308, 1.5, 4.0]])
    for x in xs:
        x = float(x)
        assert s2l(x) == R.cfref_srgb_to_linear(x)
        assert l2s(x) == R.cfref_linear_to_srgb(x)


def test_color_function_known_values():
    s2l, l2s = O.color_fns()
    assert s2l(0.0) == 0.0 and l2s(0.0) == 0.0
    assert abs(s2l(1.0) - 1.0) < 1e-15 and abs(l2s(1.0) - 1.0) < 1e-15
    assert s2l(0.04045) == 0.04045/12.92                      # the linear toe, Color.h:226-227
    assert l2s(0.0031308) == 0.0031308*12.92
    assert abs(s2l(0.5) - 0.21404114048223255) < 1e-15        # ((0.5+0.055)/1.055)^2.4
    for x in (0.001, 0.2, 0.7):
        assert abs(l2s(s2l(x)) - x) < 1e-12


def _box_py(src, dw, dh):
    """Image.cpp:1393-1447 transcribed with Python floats (= C doubles)."""
    sh, sw = src.shape[:2]
    isx, isy = sw/dw, sh/dh
    ox, oy = max(isx, 1.0), max(isy, 1.0)
    fx, fy = 1.0/ox, 1.0/oy
    ox *= 0.5
    oy *= 0.5
    out = np.zeros((dh, dw, 4), np.float32)
    for y in range(dh):
        cy = (y + 0.5)*isy
        top, bottom = max(int(cy - oy + 0.5), 0), min(int(cy + oy + 0.5), sh)
        for x in range(dw):
            cx = (x + 0.5)*isx
            left, right = max(int(cx - ox + 0.5), 0), min(int(cx + ox + 0.5), sw)
            acc, n = [0.0]*4, 0
            for i in range(top, bottom):
                if abs(i + 0.5 - cy)*fy > 0.5:
                    continue
                for j in range(left, right):
                    if abs(j + 0.5 - cx)*fx > 0.5:
                        continue
                    for k in range(4):
                        acc[k] += float(src[i, j, k])
                    n += 1
            out[y, x] = [np.float32(a/n) for a in acc]
    return out


@pytest.mark.parametrize("sw,sh,dw,dh", [(16, 16, 8, 8), (10, 6, 5, 3), (7, 5, 3, 2), (9, 1, 4, 1),
                                         (1, 8, 1, 4), (5, 5, 2, 2), (3, 3, 1, 1), (8, 8, 3, 5)])
def test_box_resize_matches_the_transcribed_loops(sw, sh, dw, dh):
    rng = np.random.default_rng(sw*100 + sh)
    src = rng.random((sh, sw, 4)).astype(np.float32)
    got = O.resize_rgbaf(src, dw, dh, filter=0 | O.FILTER_FALLBACK, color_space=0)
    assert np.array_equal(got, _box_py(src, dw, dh))


def test_box_halving_is_the_2x2_mean_in_double():
    rng = np.random.default_rng(9)
    src = rng.random((32, 48, 4)).astype(np.float32)
    got = O.resize_rgbaf(src, 24, 16, filter=0 | O.FILTER_FALLBACK)
    s = src.astype(np.float64)
    ref = ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2])/4.0).astype(np.float32)
    assert np.array_equal(got, ref)
    # the stock path (FreeImage FILTER_BOX): two passes of pair means with a FLOAT image between them
    got = O.resize_rgbaf(src, 24, 16, filter=0)
    hor = (0.5*s[:, 0::2] + 0.5*s[:, 1::2]).astype(np.float32).astype(np.float64)
    ref2 = (0.5*hor[0::2] + 0.5*hor[1::2]).astype(np.float32)
    assert np.array_equal(got, ref2)
    assert np.abs(ref2.astype(np.float64) - ref).max() < 1e-7 and not np.array_equal(ref, ref2)


def test_linear_filter_properties():
    const = np.full((12, 20, 4), 0.375, np.float32)
    for filt in (1, 1 | O.FILTER_FALLBACK):
        assert np.array_equal(O.resize_rgbaf(const, 10, 6, filter=filt), np.full((6, 10, 4), 0.375, np.float32))
        # horizontal ramp stays a ramp under the triangle filter (interior texels)
        ramp = np.tile(np.linspace(0, 1, 32, dtype=np.float32)[None, :, None], (8, 1, 4))
        out = O.resize_rgbaf(ramp, 16, 4, filter=filt)
        d = np.diff(out[0, 2:-2, 0].astype(np.float64))
        assert np.allclose(d, d[0], rtol=0, atol=1e-6)


def test_srgb_round_trip_and_alpha_untouched():
    rng = np.random.default_rng(3)
    src = rng.random((16, 16, 4)).astype(np.float32)
    lin = O.resize_rgbaf(src, 8, 8, filter=O.FILTER_FALLBACK, color_space=0)
    srgb = O.resize_rgbaf(src, 8, 8, filter=O.FILTER_FALLBACK, color_space=1)
    assert np.array_equal(lin[..., 3], srgb[..., 3])           # alpha is averaged as stored
    assert np.all(srgb[..., :3] >= lin[..., :3] - 1e-6)        # mean in linear light is brighter (convexity)
    # explicit model of Image.cpp:1337-1346 on one texel
    s2l, l2s = O.color_fns()
    blk = src[0:2, 0:2, 0]
    lin4 = [np.float32(s2l(float(v))) for v in (blk[0, 0], blk[0, 1], blk[1, 0], blk[1, 1])]
    m = np.float32((float(lin4[0]) + float(lin4[1]) + float(lin4[2]) + float(lin4[3]))/4)
    assert srgb[0, 0, 0] == np.float32(l2s(float(m)))


def test_chain_and_errors():
    img = (np.arange(64*32*4) % 251).astype(np.uint8).reshape(32, 64, 4)
    chain = O.mip_chain(img, 7)
    assert [c.shape[:2] for c in chain] == [(32, 64), (16, 32), (8, 16), (4, 8), (2, 4), (1, 2), (1, 1)]
    assert chain[0].dtype == np.float32 and chain[0][0, 0, 1] == np.float32(1/255.0)
    with pytest.raises(RuntimeError):
        O.resize_rgbaf(chain[0], 8, 8, filter=7)               # not a ResizeFilter
    with pytest.raises(RuntimeError):
        O.resize_rgbaf(chain[0], 8, 8, filter=3 | O.FILTER_FALLBACK)   # the fallback has Box and Linear only


def _fi_py(src, dw, dh, filt):
    """FreeImage's CWeightsTable + horizontal / vertical filter passes for float images, written
    independently of oracle/mipgen.c with Python floats (= C doubles); box and tent kernels."""
    W = {0: 0.5, 1: 1.0}[filt]
    F = {0: (lambda v: 1.0 if abs(v) <= 0.5 else 0.0), 1: (lambda v: 1.0 - abs(v) if abs(v) < 1.0 else 0.0)}[filt]

    def table(dst_n, src_n):
        scale = dst_n/src_n
        width, fscale = (W/scale, scale) if scale < 1.0 else (W, 1.0)
        rows = []
        for u in range(dst_n):
            center = u/scale + 0.5/scale
            left, right = max(0, int(center - width + 0.5)), min(int(center + width + 0.5), src_n)
            w = [fscale*F(fscale*(i + 0.5 - center)) for i in range(left, right)]
            tot = sum(w)
            if tot > 0 and tot != 1:
                w = [x/tot for x in w]
            rows.append((left, w))
        return rows

    def along_x(img, dst_n):
        h, n = img.shape[:2]
        if dst_n == n:
            return img
        out = np.zeros((h, dst_n, 4), np.float32)
        for u, (left, w) in enumerate(table(dst_n, n)):
            for y in range(h):
                acc = [0.0]*4
                for k, wk in enumerate(w):
                    for c in range(4):
                        acc[c] += wk*float(img[y, left + k, c])
                out[y, u] = [np.float32(a) for a in acc]
        return out
    sh, sw = src.shape[:2]
    if dw*sh <= dh*sw:
        tmp = along_x(src, dw)
        return along_x(tmp.transpose(1, 0, 2), dh).transpose(1, 0, 2)
    tmp = along_x(src.transpose(1, 0, 2), dh).transpose(1, 0, 2)
    return along_x(tmp, dw)


@pytest.mark.parametrize("filt", [0, 1])
@pytest.mark.parametrize("sw,sh,dw,dh", [(16, 16, 8, 8), (10, 6, 5, 3), (7, 5, 3, 2), (9, 1, 4, 1), (1, 8, 1, 4),
                                         (5, 5, 2, 2), (3, 3, 1, 1), (8, 8, 3, 5), (4, 4, 9, 7), (6, 3, 6, 1)])
def test_stock_box_and_linear_follow_the_freeimage_weights_table(filt, sw, sh, dw, dh):
    """Image.cpp:1348-1380: Box -> FILTER_BOX, Linear -> FILTER_BILINEAR of FreeImage_Rescale."""
    rng = np.random.default_rng(sw*100 + sh + filt)
    src = rng.random((sh, sw, 4)).astype(np.float32)
    got = O.resize_rgbaf(src, dw, dh, filter=filt, color_space=0)
    assert np.array_equal(got, _fi_py(src, dw, dh, filt))


@pytest.mark.parametrize("filt", [0, 1, 2, 3, 4])
def test_freeimage_style_filters_properties(filt):
    """All five filters through FreeImage_Rescale's algorithm (restated, parity unpinned):
    normalised weights keep constants, the interior of a ramp stays a ramp, the kernel is
    symmetric (mirrored input -> mirrored output), and the separable passes commute with a
    transpose."""
    const = np.full((20, 28, 4), 0.625, np.float32)
    assert np.allclose(O.resize_rgbaf(const, 14, 10, filter=filt), 0.625, rtol=0, atol=1e-7)
    ramp = np.tile(np.linspace(0, 1, 64, dtype=np.float32)[None, :, None], (8, 1, 4))
    d = np.diff(O.resize_rgbaf(ramp, 32, 4, filter=filt)[0, 4:-4, 0].astype(np.float64))
    assert np.allclose(d, d[0], rtol=0, atol=2e-7)
    rng = np.random.default_rng(filt)
    img = rng.random((24, 40, 4)).astype(np.float32)
    a = O.resize_rgbaf(img, 20, 12, filter=filt)
    b = O.resize_rgbaf(img[:, ::-1].copy(), 20, 12, filter=filt)[:, ::-1]
    assert np.allclose(a, b, rtol=0, atol=1e-6)
    t = O.resize_rgbaf(np.ascontiguousarray(img.transpose(1, 0, 2)), 12, 20, filter=filt).transpose(1, 0, 2)
    assert np.allclose(a, t, rtol=0, atol=1e-6)


def test_catmull_rom_halving_weights_known_answer():
    """2:1 minification: width 4, fscale 1/2, taps at +-0.25, +-0.75, +-1.25, +-1.75 of the
    stretched kernel -> weights (before normalisation) F(x)/2 with the Catmull-Rom polynomial."""
    def F(v):
        v = abs(v)
        if v < 1: return 0.5*(2 + v*v*(-5 + 3*v))
        if v < 2: return 0.5*(4 + v*(-8 + v*(5 - v)))
        return 0.0
    w = np.array([F(x) for x in (-1.75, -1.25, -0.75, -0.25, 0.25, 0.75, 1.25, 1.75)])
    w /= w.sum()
    row = np.zeros((1, 32, 4), np.float32)
    row[0, :, 0] = np.random.default_rng(1).random(32).astype(np.float32)
    out = O.resize_rgbaf(row, 16, 1, filter=3)
    u = 6                                                      # centre 12.999..: taps 9..16
    ref = float(np.dot(w, row[0, 2*u - 3:2*u + 5, 0].astype(np.float64)))
    assert abs(out[0, u, 0] - ref) < 1e-6


def test_colour_functions_reproduce_the_reference_known_answers():
    """ColorTest.SRGBConversion (lib/test/ImageTest.cpp:140-153): the reference's own golden
    values for linearToSRGB / sRGBToLinear, to EXPECT_DOUBLE_EQ's 4 ulps."""
    to_lin, to_srgb = O.color_fns()
    for x, want in ((0.0, 0.0), (0.01, 0.0998528227341283), (0.25, 0.537098730483194),
                    (0.75, 0.8808250210903), (1.0, 1.0)):
        got = to_srgb(x)
        assert abs(got - want) <= 4*np.spacing(want) + 1e-15, (x, got, want)
    for x, want in ((0.0, 0.0), (0.01, 0.000773993808049536), (0.25, 0.0508760881715568),
                    (0.75, 0.522521553968392), (1.0, 1.0)):
        got = to_lin(x)
        assert abs(got - want) <= 4*np.spacing(want) + 1e-15, (x, got, want)
