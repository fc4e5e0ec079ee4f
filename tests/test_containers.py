"""Container writers: DDS-DX10 / KTX1 serialisation of the payloads.

The DDS files are opened with Pillow -- an independent reader AND an independent BCn decoder --
and the pixels must equal the oracle decoder's on the same payload: this pins the whole chain
payload layout -> container -> third-party decode.  Mirrors lib/test/TextureSaveTest.cpp:77-266
(16x16 black image saved in every legal (format, type); illegal pairs are Unsupported).
"""
import io
import struct

import numpy as np
import pytest

import oracle_lib as O
from cuttlefish_amd import ColorSpace, Format, Type, payload_size, synth
from cuttlefish_amd import containers as C


def _mips(img, levels):
    out = [img]
    for _ in range(1, levels):
        a = out[-1].astype(np.uint16)
        h, w = a.shape[:2]
        a = a[:h // 2 * 2, :w // 2 * 2]
        out.append(((a[0::2, 0::2] + a[1::2, 0::2] + a[0::2, 1::2] + a[1::2, 1::2] + 2) // 4)
                   .astype(np.uint8))
    return out


@pytest.mark.parametrize("fmt,typ", [(Format.BC1_RGB, Type.UNorm), (Format.BC2, Type.UNorm),
                                     (Format.BC3, Type.UNorm), (Format.BC4, Type.UNorm),
                                     (Format.BC5, Type.UNorm), (Format.BC7, Type.UNorm)])
def test_dds_opens_in_pillow_and_matches_oracle_decode(fmt, typ):
    PIL = pytest.importorskip("PIL.Image")
    img = synth.photo(64, 64, seed=3)
    levels = [O.encode(m, int(fmt), int(typ), quality=1) for m in _mips(img, 5)]   # 64 .. 4
    buf = io.BytesIO()
    n = C.write_dds(buf, fmt, typ, 64, 64, levels)
    data = buf.getvalue()
    assert n == len(data) == 148 + sum(l.nbytes for l in levels)
    info = C.read_dds(data)
    # arraySize = Texture::depth() = 0 for a texture that is not an array (SaveDds.cpp:637); Pillow still opens it
    assert (info["width"], info["height"], info["levels"], info["elements"]) == (64, 64, 5, 0)
    assert data[info["offset"]:info["offset"] + levels[0].nbytes] == levels[0].tobytes()
    im = PIL.open(io.BytesIO(data))
    im.load()
    got = np.asarray(im)
    ref = O.decode(levels[0], int(fmt), 64, 64, int(typ))
    if got.ndim == 2:
        assert np.array_equal(got, ref[..., 0])
    elif fmt == Format.BC5:
        assert np.array_equal(got[..., :2], ref[..., :2])
    else:
        c = got.shape[2]
        assert np.array_equal(got, ref[..., :c])


def test_dds_header_fields_follow_savedds():
    black = np.zeros((16, 16, 4), np.uint8)
    black[..., 3] = 255                       # TextureSaveTest.cpp:83-91
    pay = O.encode(black, int(Format.BC7), 0, quality=0)
    buf = io.BytesIO()
    C.write_dds(buf, Format.BC7, Type.UNorm, 16, 16, [pay], color_space=ColorSpace.sRGB)
    d = buf.getvalue()
    magic, size, flags, h, w, pitch, depth, mips = struct.unpack_from("<8I", d, 0)
    assert magic == 0x20534444 and size == 124
    assert flags == (0x1 | 0x2 | 0x4 | 0x1000 | 0x20000 | 0x8)   # Required | MipmapCount | Pitch
    assert (h, w, depth, mips) == (16, 16, 0, 1)
    assert pitch == 4 * 16                                        # blocks per row * block size
    pf_size, pf_flags = struct.unpack_from("<2I", d, 4 + 72)
    assert (pf_size, pf_flags, d[4 + 80:4 + 84]) == (32, 0x4, b"DX10")
    assert struct.unpack_from("<I", d, 4 + 104)[0] == 0x1000      # caps: texture only
    dxgi, dim, misc, array, misc2 = struct.unpack_from("<5I", d, 128)
    assert (dxgi, dim, misc, array, misc2) == (99, 3, 0, 0, 1)    # BC7_UNORM_SRGB, 2-D, arraySize = depth() = 0, straight alpha
    assert len(d) == 148 + 256


def test_dds_unsupported_pairs_and_size_checks():
    with pytest.raises(ValueError):
        C.write_dds(io.BytesIO(), Format.ETC1, Type.UNorm, 16, 16, [b"\0" * 128])   # no DXGI format
    with pytest.raises(ValueError):
        C.write_dds(io.BytesIO(), Format.BC1_RGB, Type.UNorm, 16, 16, [b"\0" * 100])


def test_dds_array_order_is_element_then_mip():
    a = [bytes([1]) * 128, bytes([2]) * 32]     # 16x16 BC1 + 8x8 mip
    b = [bytes([3]) * 128, bytes([4]) * 32]
    buf = io.BytesIO()
    C.write_dds(buf, Format.BC1_RGB, Type.UNorm, 16, 16, [a, b])
    d = buf.getvalue()
    info = C.read_dds(d)
    assert info["elements"] == 2 and info["levels"] == 2
    body = d[info["offset"]:]
    assert body == a[0] + a[1] + b[0] + b[1]
    assert struct.unpack_from("<I", d, 4 + 104)[0] == (0x1000 | 0x400000 | 0x8)   # texture|mipmap|complex


@pytest.mark.parametrize("fmt,typ,internal,base", [
    (Format.BC7, Type.UNorm, 0x8E8C, 0x1908), (Format.ETC2_R8G8B8A8, Type.UNorm, 0x9278, 0x1908),
    (Format.ASTC_6x6, Type.UNorm, 0x93B4, 0x1908), (Format.EAC_R11, Type.SNorm, 0x9271, 0x1903),
    (Format.BC6H, Type.UFloat, 0x8E8F, 0x1907)])
def test_ktx_layout_follows_savektx(fmt, typ, internal, base):
    w, h = 24, 16
    dims = C.mip_dims(w, h, 3)
    levels = [bytes([i + 1]) * payload_size(fmt, typ, dw, dh) for i, (dw, dh) in enumerate(dims)]
    buf = io.BytesIO()
    n = C.write_ktx(buf, fmt, typ, w, h, levels)
    d = buf.getvalue()
    assert n == len(d) == 64 + sum(4 + len(l) for l in levels)
    assert d[:12] == bytes([0xAB, 0x4B, 0x54, 0x58, 0x20, 0x31, 0x31, 0xBB, 0x0D, 0x0A, 0x1A, 0x0A])
    f = struct.unpack_from("<13I", d, 12)
    assert f[0] == 0x04030201
    assert f[1:6] == (0, 1, 0, internal, base)            # type, typeSize, format, internal, base
    assert f[6:13] == (w, h, 0, 0, 1, 3, 0)               # w, h, depth, array, faces, mips, kv bytes
    off = 64
    for l in levels:
        assert struct.unpack_from("<I", d, off)[0] == len(l)
        assert d[off + 4:off + 4 + len(l)] == l
        off += 4 + len(l)


def test_ktx_srgb_and_array():
    lv = [bytes(16 * 4)]                                   # one 8x8 BC7 level per element
    buf = io.BytesIO()
    C.write_ktx(buf, Format.BC7, Type.UNorm, 8, 8, [lv, lv, lv], color_space=ColorSpace.sRGB)
    d = buf.getvalue()
    f = struct.unpack_from("<13I", d, 12)
    assert f[4] == 0x8E8D and f[9] == 3                    # sRGB BPTC, 3 array elements
    assert struct.unpack_from("<I", d, 64)[0] == 3 * 64    # imageSize covers all elements


# ---- PVR v3 and the uncompressed formats ------------------------------------------------------

def test_pvr_header_metadata_and_surface_order_follow_savepvr():
    img = synth.photo(32, 16, seed=2)
    e0 = [O.encode(m, int(Format.BC1_RGBA), 0, quality=0) for m in _mips(img, 3)]
    e1 = [O.encode(m[::-1].copy(), int(Format.BC1_RGBA), 0, quality=0) for m in _mips(img, 3)]
    buf = io.BytesIO()
    n = C.write_pvr(buf, Format.BC1_RGBA, Type.UNorm, 32, 16, [e0, e1], color_space=ColorSpace.sRGB)
    data = buf.getvalue()
    assert n == len(data)
    h = C.read_pvr(data)
    assert data[:4] == b"PVR\x03"
    assert (h["width"], h["height"], h["depth"], h["elements"], h["faces"], h["levels"]) == (32, 16, 1, 2, 1, 3)
    assert h["pixel_format"] == 7 and h["channel_type"] == 0 and h["color_space"] == 1 and h["flags"] == 0
    # two CTFS blocks: BC1A marker, then the array marker (SavePvr.cpp:526-560)
    assert h["metadata"] == (b"CTFSBC1A" + struct.pack("<II", 4, 0) + b"CTFSARRY" + struct.pack("<II", 4, 0))
    body = data[h["offset"]:]
    want = b"".join(bytes(e[l]) for l in range(3) for e in (e0, e1))      # mip -> element
    assert body == want
    buf = io.BytesIO()
    C.write_pvr(buf, Format.BC7, Type.UNorm, 32, 16, e0[:1]*0 + [O.encode(img, int(Format.BC7), 0, quality=0)])
    h = C.read_pvr(buf.getvalue())
    assert h["metadata"] == b"" and h["offset"] == 52 and h["pixel_format"] == 15


def test_pvr_pixel_formats_and_channel_types():
    g = C.pvr_pixel_format
    assert g(Format.R5G6B5) == (ord("r") | ord("g") << 8 | ord("b") << 16 | 5 << 32 | 6 << 40 | 5 << 48)
    assert g(Format.A2B10G10R10) == (ord("a") | ord("b") << 8 | ord("g") << 16 | ord("r") << 24 |
                                      2 << 32 | 10 << 40 | 10 << 48 | 10 << 56)
    assert g(Format.R8) == (ord("r") | 8 << 32)
    assert g(Format.E5B9G9R9_UFloat) == 19 and g(Format.ETC1) == 6 and g(Format.ASTC_4x4) == 27
    assert g(Format.ASTC_12x12) == 40 and g(Format.EAC_R11G11) == 26 and g(Format.ETC2_R8G8B8A1) == 24
    assert g(Format.BC2, alpha=2) == 8 and g(Format.BC3, alpha=2) == 10 and g(Format.BC3) == 11
    t = C.pvr_channel_type
    assert [t(Format.R8G8B8A8, k) for k in (Type.UNorm, Type.SNorm, Type.UInt, Type.Int)] == [0, 1, 2, 3]
    assert [t(Format.R16G16, k) for k in (Type.UNorm, Type.SNorm, Type.UInt, Type.Int, Type.Float)] == [4, 5, 6, 7, 12]
    assert [t(Format.R32, k) for k in (Type.UInt, Type.Int, Type.Float)] == [10, 11, 12]
    assert t(Format.A2R10G10B10, Type.UNorm) == 8 and t(Format.R5G6B5, Type.UNorm) == 4
    assert t(Format.BC5, Type.SNorm) == 1 and t(Format.EAC_R11, Type.UNorm) == 4 and t(Format.BC7, Type.UNorm) == 0
    assert t(Format.BC6H, Type.UFloat) == 13 and t(Format.BC6H, Type.Float) == 12
    assert t(Format.B10G11R11_UFloat, Type.UFloat) == 13


def test_uncompressed_dds_opens_in_pillow():
    PIL = pytest.importorskip("PIL.Image")
    img = synth.photo(40, 24, seed=8)
    payload = O.std_pack(img, int(Format.R8G8B8A8), int(Type.UNorm))
    buf = io.BytesIO()
    C.write_dds(buf, Format.R8G8B8A8, Type.UNorm, 40, 24, [payload])
    hdr = C.read_dds(buf.getvalue())
    assert hdr["dxgi"] == 28 and hdr["pitch"] == 160 and hdr["alpha_mode"] == 1
    got = np.asarray(PIL.open(io.BytesIO(buf.getvalue())).convert("RGBA"))
    assert np.array_equal(got, img)
    for fmt, typ, dxgi in ((Format.R16G16B16A16, Type.Float, 10), (Format.R32G32B32, Type.Float, 6),
                           (Format.R5G6B5, Type.UNorm, 85), (Format.E5B9G9R9_UFloat, Type.UFloat, 67),
                           (Format.B10G11R11_UFloat, Type.UFloat, 26), (Format.R8, Type.SNorm, 63)):
        buf = io.BytesIO()
        C.write_dds(buf, fmt, typ, 40, 24, [O.std_pack(img, int(fmt), int(typ))])
        assert C.read_dds(buf.getvalue())["dxgi"] == dxgi
    for fmt, typ in ((Format.R8G8B8, Type.UNorm), (Format.B8G8R8, Type.UNorm), (Format.R4G4B4A4, Type.UNorm)):
        with pytest.raises(ValueError):                      # getDdsFormat has no entry: Unsupported
            C.write_dds(io.BytesIO(), fmt, typ, 40, 24, [O.std_pack(img, int(fmt), int(typ))])


def test_ktx_uncompressed_formats_follow_getformatinfo_and_pad_rows():
    img = synth.photo(5, 3, seed=4)                      # 5 pixels: 15-byte RGB8 rows -> 1 byte of padding
    payload = O.std_pack(img, int(Format.R8G8B8), int(Type.UNorm))
    buf = io.BytesIO()
    n = C.write_ktx(buf, Format.R8G8B8, Type.UNorm, 5, 3, [payload], color_space=ColorSpace.sRGB)
    data = buf.getvalue()
    assert n == len(data) == 64 + 4 + 3*16
    gl_type, type_size, gl_format, internal, base = struct.unpack_from("<5I", data, 16)
    assert (gl_type, type_size, gl_format, internal, base) == (0x1401, 1, 0x1907, 0x8C41, 0x1907)
    assert struct.unpack_from("<I", data, 64)[0] == 48
    rows = np.frombuffer(data, np.uint8, offset=68).reshape(3, 16)
    assert np.array_equal(rows[:, :15].reshape(3, 5, 3), img[..., :3]) and (rows[:, 15] == 0).all()
    for fmt, typ, want in ((Format.R5G6B5, Type.UNorm, (0x8363, 2, 0x1907, 0x8D62, 0x1907)),
                           (Format.R8, Type.SNorm, (0x1400, 1, 0x1903, 0x8F94, 0x1909)),
                           (Format.R8G8, Type.SNorm, (0x1401, 1, 0x8227, 0x8F95, 0x190A)),
                           (Format.R8G8B8A8, Type.UInt, (0x1401, 1, 0x8D99, 0x8D7C, 0x1908)),
                           (Format.A2R10G10B10, Type.UInt, (0x8368, 4, 0x8D9B, 0x906F, 0x80E1)),
                           (Format.R16G16B16A16, Type.Float, (0x140B, 2, 0x1908, 0x881A, 0x1908)),
                           (Format.R32G32, Type.Int, (0x1404, 4, 0x8227, 0x823B, 0x190A)),
                           (Format.E5B9G9R9_UFloat, Type.UFloat, (0x8C3E, 4, 0x1907, 0x8C3D, 0x1907))):
        buf = io.BytesIO()
        C.write_ktx(buf, fmt, typ, 5, 3, [O.std_pack(img, int(fmt), int(typ))])
        assert struct.unpack_from("<5I", buf.getvalue(), 16) == want, fmt
    for fmt in (Format.R4G4, Format.A4R4G4B4, Format.B8G8R8):                 # SaveKtx.cpp:1174-1176
        with pytest.raises(ValueError):
            C.write_ktx(io.BytesIO(), fmt, Type.UNorm, 5, 3, [O.std_pack(img, int(fmt), 0)])


def test_writers_accept_exactly_what_the_reference_save_tests_expect():
    """lib/test/TextureSaveTest.cpp:268-700 lists, per container, which (format, type) pairs save
    and which are Unsupported (fixture: tests/golden/save_expectations.json, extracted by
    tests/golden/make_save_expectations.py).  The same 16x16 texture: header sizes 148 / 68 / 52
    bytes + payload (TextureSaveTest.cpp:252-265)."""
    import json
    import os
    from cuttlefish_amd import api
    exp = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "save_expectations.json")))
    writers = {"DDS": (C.write_dds, 148), "KTX": (C.write_ktx, 68), "PVR": (C.write_pvr, 52)}
    checked = 0
    for kind, (writer, header) in writers.items():
        for key, ok in exp[kind].items():
            fname, tname = key.split("/")
            if fname.startswith("PVRTC"):
                continue                                   # closed-source codec: not in this backend
            fmt, typ = getattr(Format, fname), getattr(Type, tname)
            size = payload_size(fmt, typ, 16, 16)
            buf = io.BytesIO()
            try:
                n = writer(buf, fmt, typ, 16, 16, [bytes(size)])
                got = True
            except ValueError:
                got = False
            assert got == ok, (kind, key)
            if ok:
                extra = 16 if (kind == "PVR" and fname.startswith("BC1")) else 0   # the BC1/BC1A metadata block
                assert n == header + extra + size, (kind, key, n)
            checked += 1
    assert checked > 250


# ---- cube maps, cube arrays, 3-D textures (general [level][depth][face] writers) -------------

def _tagged(fmt, typ, w, h, tag):
    return bytes([tag % 251])*payload_size(fmt, typ, w, h)


def _layout(dimension, depth, levels, w=16, h=16, fmt=None, typ=None, is_array=None):
    from cuttlefish_amd import Format, Type
    fmt, typ = fmt or Format.BC1_RGB, typ or Type.UNorm
    faces = 6 if dimension == "cube" else 1
    surf, tag = [], 0
    for l in range(levels):
        nd = max(depth >> l, 1) if dimension == "3d" else max(depth, 1)
        lvl = []
        for d in range(nd):
            lvl.append([_tagged(fmt, typ, max(1, w >> l), max(1, h >> l), 16*l + 4*d + f + 1) for f in range(faces)])
        surf.append(lvl)
    return C.TextureLayout(fmt, typ, w, h, surf, dimension=dimension, depth=depth, is_array=is_array)


def _runs(payload):
    """the distinct tag bytes of a payload, in order"""
    out = []
    for b in payload:
        if not out or out[-1] != b:
            out.append(b)
    return out


def test_cube_map_orders_follow_the_three_writers():
    tex = _layout("cube", 0, 2)
    tag = lambda l, d, f: 16*l + 4*d + f + 1
    buf = io.BytesIO(); C.write_dds_texture(buf, tex); data = buf.getvalue()
    hdr = C.read_dds(data)
    caps, caps2 = struct.unpack_from("<2I", data, 4 + 104)
    dxgi, dim, misc, array, _ = struct.unpack_from("<5I", data, 4 + 124)
    assert caps2 == 0xFE00 and misc == 0x4 and dim == 3 and array == 0        # SaveDds.cpp:600-604, :624-627, :637
    assert caps & 0x8                                                         # mip levels -> complex
    # DDS: face -> level (SaveDds.cpp:657-680)
    assert _runs(data[hdr["offset"]:]) == [tag(l, 0, f) for f in range(6) for l in range(2)]
    buf = io.BytesIO(); C.write_ktx_texture(buf, tex); k = buf.getvalue()
    w, h, d, arr, faces, levels, kv = struct.unpack_from("<7I", k, 12 + 4 + 20)
    assert (d, arr, faces, levels) == (0, 0, 6, 2)
    size0 = struct.unpack_from("<I", k, 64)[0]
    assert size0 == len(tex.surfaces[0][0][0])                                # ONE face for a non-array cube map
    # KTX: level -> face
    body = k[64:]
    assert _runs(body[4:4 + 6*size0]) == [tag(0, 0, f) for f in range(6)]
    buf = io.BytesIO(); C.write_pvr_texture(buf, tex); p = buf.getvalue()
    ph = C.read_pvr(p)
    assert (ph["depth"], ph["elements"], ph["faces"], ph["levels"]) == (1, 1, 6, 2)
    assert _runs(p[ph["offset"]:]) == [tag(l, 0, f) for l in range(2) for f in range(6)]


def test_3d_texture_headers_and_slice_order():
    tex = _layout("3d", 4, 3)                          # 16x16x4, levels with 4 / 2 / 1 slices
    tag = lambda l, d: 16*l + 4*d + 1
    buf = io.BytesIO(); C.write_dds_texture(buf, tex); data = buf.getvalue()
    _, size, flags, height, width, pitch, depth, levels = struct.unpack_from("<8I", data, 0)
    caps, caps2 = struct.unpack_from("<2I", data, 4 + 104)
    dxgi, dim, misc, array, _ = struct.unpack_from("<5I", data, 4 + 124)
    assert flags & 0x800000 and depth == 4 and caps2 == 0x200000 and dim == 4 and array == 1 and caps & 0x8
    assert _runs(data[4 + 124 + 20:]) == [tag(l, d) for l in range(3) for d in range(max(4 >> l, 1))]
    buf = io.BytesIO(); C.write_ktx_texture(buf, tex); k = buf.getvalue()
    w, h, d, arr, faces, levels, kv = struct.unpack_from("<7I", k, 12 + 4 + 20)
    assert (d, arr, faces, levels) == (4, 0, 1, 3)
    assert struct.unpack_from("<I", k, 64)[0] == 4*len(tex.surfaces[0][0][0])
    buf = io.BytesIO(); C.write_pvr_texture(buf, tex); p = buf.getvalue()
    ph = C.read_pvr(p)
    assert (ph["depth"], ph["elements"], ph["faces"]) == (4, 1, 1)
    assert _runs(p[ph["offset"]:]) == [tag(l, d) for l in range(3) for d in range(max(4 >> l, 1))]


def test_cube_array_and_equivalence_with_the_2d_writers():
    tex = _layout("cube", 2, 1, is_array=True)         # two cubes
    tag = lambda d, f: 4*d + f + 1
    buf = io.BytesIO(); C.write_dds_texture(buf, tex); data = buf.getvalue()
    dxgi, dim, misc, array, _ = struct.unpack_from("<5I", data, 4 + 124)
    assert array == 2 and misc == 0x4
    assert _runs(data[4 + 124 + 20:]) == [tag(d, f) for d in range(2) for f in range(6)]
    buf = io.BytesIO(); C.write_ktx_texture(buf, tex); k = buf.getvalue()
    assert struct.unpack_from("<I", k, 64)[0] == 2*6*len(tex.surfaces[0][0][0])    # arrays: every face counted
    # a plain 2-D array through the general writer = the 2-D writer, byte for byte
    from cuttlefish_amd import Format, Type, payload_size
    lv = [[bytes([7 + e + 3*l])*payload_size(Format.BC7, Type.UNorm, 8 >> l, 8 >> l) for l in range(2)] for e in range(3)]
    gen = C.TextureLayout(Format.BC7, Type.UNorm, 8, 8, [[[lv[e][l]] for e in range(3)] for l in range(2)], "2d", depth=3)
    for wa, wb in ((C.write_dds, C.write_dds_texture), (C.write_ktx, C.write_ktx_texture), (C.write_pvr, C.write_pvr_texture)):
        a, b = io.BytesIO(), io.BytesIO()
        wa(a, Format.BC7, Type.UNorm, 8, 8, lv)
        wb(b, gen)
        assert a.getvalue() == b.getvalue(), wa.__name__
