"""The uncompressed ("standard") converters on the GPU through the
C ABI: byte-identical to oracle/std_pack.c for every (format, type) createConverter accepts
(Converter.cpp:38-337), every source pixel type, ragged sizes and special float values."""
import numpy as np
import pytest

import oracle_lib as O
from test_oracle_stdpack import ALL_PAIRS, F, FLOAT, UFLOAT, UNORM
from cuttlefish_amd import Context, Format, Type, make_params, payload_size, query

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    with Context(0) as c:
        yield c


def float_image(h, w, seed, span):
    rng = np.random.default_rng(seed)
    img = (rng.random((h, w, 4)).astype(np.float32)*2 - 0.6)*np.float32(span)
    flat = img.reshape(-1)
    specials = np.array([0.0, -0.0, 0.5, -0.5, 1.5, 2.5, -2.5, 0.5/255, 1.5/255, 0.5/15, 0.5/31, 0.5/63,
                         0.5/1023, 127.5, -128.5, 32767.5, 65535.5, 1e-45, -1e-45, 1e-39, 6e-8, 65504.0,
                         65520.0, 1e9, -1e9, 4294967296.0, 2147483648.0, -2147483648.0, 3e38, -3e38,
                         np.inf, -np.inf, np.nan], np.float32)
    flat[:specials.size] = specials
    return img


@pytest.mark.parametrize("fmt,typ", ALL_PAIRS)
def test_every_legal_pair_matches_the_oracle_from_float_sources(ctx, fmt, typ):
    span = 1.0 if typ <= 1 or typ == FLOAT else (70000.0 if fmt < F["R32"] else 3e9)
    if typ == UFLOAT:
        span = 40000.0
    img = float_image(37, 61, seed=fmt*8 + typ, span=span)     # 2257 pixels: ragged last workgroup
    if fmt >= F["R16"] and fmt <= F["R16G16B16A16"] and typ == FLOAT:
        img = np.where(np.isnan(img), np.float32(7.0), img)     # NaN payloads: see DESIGN, not compared
    p = make_params(Format(fmt), Type(typ), 2)
    got = ctx.encode([img], p)[0]
    assert got.size == payload_size(Format(fmt), Type(typ), 61, 37) == 37*61*O.std_pixel_bytes(fmt, typ)
    assert query(Format(fmt), Type(typ)) == (1, 1, O.std_pixel_bytes(fmt, typ))
    assert np.array_equal(got, O.std_pack(img, fmt, typ))


@pytest.mark.parametrize("fmt,typ", [(F["R8G8B8A8"], 0), (F["B8G8R8"], 0), (F["R5G6B5"], 0), (F["R4G4"], 0),
                                     (F["R16G16B16"], 0), (F["R16G16B16A16"], FLOAT), (F["R32G32B32"], FLOAT),
                                     (F["A2B10G10R10"], 0), (F["E5B9G9R9"], UFLOAT), (F["B10G11R11"], UFLOAT),
                                     (F["R8"], 1), (F["R32G32B32A32"], 2)])
@pytest.mark.parametrize("src", ["u8", "f16"])
def test_rgba8_and_half_sources(ctx, fmt, typ, src):
    rng = np.random.default_rng(fmt + 100)
    if src == "u8":
        img = rng.integers(0, 256, size=(19, 83, 4), dtype=np.uint8)
    else:
        img = (rng.random((19, 83, 4))*3 - 1).astype(np.float16)
    got = ctx.encode([img], make_params(Format(fmt), Type(typ), 2))[0]
    assert np.array_equal(got, O.std_pack(img, fmt, typ))


@pytest.mark.parametrize("w,h", [(1, 1), (3, 1), (1, 1025), (1024, 1), (1025, 3), (4096, 2)])
def test_sizes_around_the_workgroup_boundary(ctx, w, h):
    rng = np.random.default_rng(w*7 + h)
    img = rng.random((h, w, 4)).astype(np.float32)
    for fmt, typ in ((F["R8"], 0), (F["R5G6B5"], 0), (F["R8G8B8"], 0), (F["R16G16B16"], 0),
                     (F["R8G8B8A8"], 0), (F["R16G16B16A16"], 0), (F["R32G32B32"], FLOAT),
                     (F["R32G32B32A32"], FLOAT)):
        got = ctx.encode([img], make_params(Format(fmt), Type(typ), 2))[0]
        assert np.array_equal(got, O.std_pack(img, fmt, typ)), (fmt, typ)


def test_many_surfaces_and_bottom_up_rows(ctx):
    rng = np.random.default_rng(1)
    imgs = [rng.random((h, w, 4)).astype(np.float32) for w, h in ((64, 64), (32, 32), (5, 3), (1, 1), (7, 130))]
    p = make_params(Format.B8G8R8, Type.UNorm, 2)
    for got, img in zip(ctx.encode(imgs, p), imgs):
        assert np.array_equal(got, O.std_pack(img, F["B8G8R8"], UNORM))
    big = rng.random((515, 257, 4)).astype(np.float32)
    storage = np.ascontiguousarray(big[::-1])
    view = storage[::-1]                                       # negative pitch: strip pipeline
    assert view.strides[0] < 0
    for fmt, typ in ((F["B8G8R8"], 0), (F["R16G16B16"], 0), (F["R32G32B32A32"], FLOAT), (F["R4G4"], 0)):
        got = ctx.encode([view], make_params(Format(fmt), Type(typ), 2))[0]
        assert np.array_equal(got, O.std_pack(big, fmt, typ)), (fmt, typ)


def test_device_surfaces_with_a_row_pitch(ctx):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(2)
    w, h, pitch_px = 301, 77, 320
    host = rng.random((h, pitch_px, 4)).astype(np.float32)
    dsrc = torch.from_numpy(host).cuda()
    for fmt, typ in ((F["R8G8B8"], 0), (F["R16G16B16A16"], FLOAT), (F["A2R10G10B10"], 0)):
        bpp = O.std_pixel_bytes(fmt, typ)
        dout = torch.zeros(w*h*bpp, dtype=torch.uint8, device="cuda")
        ctx.encode_device([dict(pixels=dsrc.data_ptr(), pixel_type=1, width=w, height=h,
                                row_pitch_bytes=pitch_px*16, out=dout.data_ptr(), out_capacity=dout.numel())],
                          make_params(Format(fmt), Type(typ), 2))
        torch.cuda.synchronize()
        assert np.array_equal(dout.cpu().numpy(), O.std_pack(host[:, :w], fmt, typ)), (fmt, typ)


def test_illegal_pairs_are_refused(ctx):
    img = np.zeros((4, 4, 4), np.float32)
    for fmt, typ in ((F["R5G6B5"], 1), (F["R32"], 0), (F["B10G11R11"], FLOAT), (F["R8"], FLOAT)):
        with pytest.raises(Exception):
            ctx.encode([img], make_params(Format(fmt), Type(typ), 2))
