"""DDS (DX10 header), KTX 1.1 and PVR v3 writers -- SURVEY section 8(f) row 2: "container writers
(DDS-DX10 / KTX1 / PVR3) ... so outputs open in standard viewers".

Pure serialisation, host side, mirrors the reference writers field for field:
  saveDds   lib/src/SaveDds.cpp:565-683   (header flags :578-598, DX10 header :604-652,
                                            surface order element -> face -> mip :657-680)
  saveKtx   lib/src/SaveKtx.cpp:1189-1290 (header :1198-1221, per level imageSize :1224-1248,
                                            then depth -> face payloads :1250-1262)
  savePvr   lib/src/SavePvr.cpp:478-600   (header :487-524, CTFS metadata :526-575,
                                            surface order mip -> depth -> face :580-595;
                                            pixel format :272-470, channel type :107-268)
All three cover the uncompressed formats of csrc/std_pack.hip too (getDdsFormat,
SaveDds.cpp:252-440; getFormatInfo, SaveKtx.cpp:200-630, rows padded to 4 bytes :1240-1285).
2-D textures, 2-D arrays, cube maps, cube arrays and 3-D textures: `write_dds` / `write_ktx` /
`write_pvr` take one 2-D texture or array; `write_dds_texture` / `write_ktx_texture` /
`write_pvr_texture` take the reference's general [level][depth][face] surface lists with a
dimension ("2d", "cube", "3d").  `read_dds` parses what the DDS writers emit (round-trip tests;
Pillow is the independent reader used by tests/test_containers.py).
"""
from __future__ import annotations

import struct
from typing import Sequence

from .api import Alpha, ColorSpace, Format, Type, payload_size, query

DDS_MAGIC = 0x20534444          # "DDS "  (SaveDds.cpp:26)
_DDSD_CAPS, _DDSD_HEIGHT, _DDSD_WIDTH, _DDSD_PITCH = 0x1, 0x2, 0x4, 0x8
_DDSD_PIXELFORMAT, _DDSD_MIPMAPCOUNT = 0x1000, 0x20000
_DDPF_FOURCC = 0x4
_DDSCAPS_COMPLEX, _DDSCAPS_MIPMAP, _DDSCAPS_TEXTURE = 0x8, 0x400000, 0x1000
_DIM_TEXTURE2D = 3
_ALPHA_MODE = {Alpha.None_: 3, Alpha.Standard: 1, Alpha.PreMultiplied: 2, Alpha.Encoded: 4}

# DXGI_FORMAT values (getDdsFormat, SaveDds.cpp:440-508): (linear, sRGB or None)
_DXGI = {
    (Format.BC1_RGB, Type.UNorm): (71, 72), (Format.BC1_RGBA, Type.UNorm): (71, 72),
    (Format.BC2, Type.UNorm): (74, 75), (Format.BC3, Type.UNorm): (77, 78),
    (Format.BC4, Type.UNorm): (80, None), (Format.BC4, Type.SNorm): (81, None),
    (Format.BC5, Type.UNorm): (83, None), (Format.BC5, Type.SNorm): (84, None),
    (Format.BC6H, Type.UFloat): (95, None), (Format.BC6H, Type.Float): (96, None),
    (Format.BC7, Type.UNorm): (98, 99),
}

# OpenGL enums (getFormatInfo, SaveKtx.cpp): (internalFormat linear, sRGB or None, base)
_GL_RED, _GL_RG, _GL_RGB, _GL_RGBA = 0x1903, 0x8227, 0x1907, 0x1908
_GL = {
    (Format.BC1_RGB, Type.UNorm): (0x83F0, 0x8C4C, _GL_RGB),
    (Format.BC1_RGBA, Type.UNorm): (0x83F1, 0x8C4D, _GL_RGBA),
    (Format.BC2, Type.UNorm): (0x83F2, 0x8C4E, _GL_RGBA),
    (Format.BC3, Type.UNorm): (0x83F3, 0x8C4F, _GL_RGBA),
    (Format.BC4, Type.UNorm): (0x8DBB, None, _GL_RED), (Format.BC4, Type.SNorm): (0x8DBC, None, _GL_RED),
    (Format.BC5, Type.UNorm): (0x8DBD, None, _GL_RG), (Format.BC5, Type.SNorm): (0x8DBE, None, _GL_RG),
    (Format.BC6H, Type.UFloat): (0x8E8F, None, _GL_RGB), (Format.BC6H, Type.Float): (0x8E8E, None, _GL_RGB),
    (Format.BC7, Type.UNorm): (0x8E8C, 0x8E8D, _GL_RGBA),
    (Format.ETC1, Type.UNorm): (0x8D64, None, _GL_RGB),
    (Format.ETC2_R8G8B8, Type.UNorm): (0x9274, 0x9275, _GL_RGB),
    (Format.ETC2_R8G8B8A1, Type.UNorm): (0x9276, 0x9277, _GL_RGBA),
    (Format.ETC2_R8G8B8A8, Type.UNorm): (0x9278, 0x9279, _GL_RGBA),
    (Format.EAC_R11, Type.UNorm): (0x9270, None, _GL_RED), (Format.EAC_R11, Type.SNorm): (0x9271, None, _GL_RED),
    (Format.EAC_R11G11, Type.UNorm): (0x9272, None, _GL_RG), (Format.EAC_R11G11, Type.SNorm): (0x9273, None, _GL_RG),
}
# uncompressed formats, getDdsFormat (SaveDds.cpp:252-440): (format, type) -> DXGI (linear, sRGB)
_U, _S, _UI, _I, _UF, _F = Type.UNorm, Type.SNorm, Type.UInt, Type.Int, Type.UFloat, Type.Float
_DXGI.update({
    (Format.R4G4, _U): (112, None), (Format.A4R4G4B4, _U): (115, None), (Format.R5G6B5, _U): (85, None),
    (Format.A1R5G5B5, _U): (86, None),
    (Format.R8, _U): (61, None), (Format.R8, _UI): (62, None), (Format.R8, _S): (63, None), (Format.R8, _I): (64, None),
    (Format.R8G8, _U): (49, None), (Format.R8G8, _UI): (50, None), (Format.R8G8, _S): (51, None), (Format.R8G8, _I): (52, None),
    (Format.R8G8B8A8, _U): (28, 29), (Format.R8G8B8A8, _UI): (30, None), (Format.R8G8B8A8, _S): (31, None),
    (Format.R8G8B8A8, _I): (32, None), (Format.B8G8R8A8, _U): (87, 91),
    (Format.A2B10G10R10, _U): (24, None), (Format.A2B10G10R10, _UI): (25, None),
    (Format.R16, _F): (54, None), (Format.R16, _U): (56, None), (Format.R16, _UI): (57, None),
    (Format.R16, _S): (58, None), (Format.R16, _I): (59, None),
    (Format.R16G16, _F): (34, None), (Format.R16G16, _U): (35, None), (Format.R16G16, _UI): (36, None),
    (Format.R16G16, _S): (37, None), (Format.R16G16, _I): (38, None),
    (Format.R16G16B16A16, _F): (10, None), (Format.R16G16B16A16, _U): (11, None), (Format.R16G16B16A16, _UI): (12, None),
    (Format.R16G16B16A16, _S): (13, None), (Format.R16G16B16A16, _I): (14, None),
    (Format.R32, _F): (41, None), (Format.R32, _UI): (42, None), (Format.R32, _I): (43, None),
    (Format.R32G32, _F): (16, None), (Format.R32G32, _UI): (17, None), (Format.R32G32, _I): (18, None),
    (Format.R32G32B32, _F): (6, None), (Format.R32G32B32, _UI): (7, None), (Format.R32G32B32, _I): (8, None),
    (Format.R32G32B32A32, _F): (2, None), (Format.R32G32B32A32, _UI): (3, None), (Format.R32G32B32A32, _I): (4, None),
    (Format.B10G11R11_UFloat, _UF): (26, None), (Format.E5B9G9R9_UFloat, _UF): (67, None),
})

_ASTC = [Format.ASTC_4x4, Format.ASTC_5x4, Format.ASTC_5x5, Format.ASTC_6x5, Format.ASTC_6x6,
         Format.ASTC_8x5, Format.ASTC_8x6, Format.ASTC_8x8, Format.ASTC_10x5, Format.ASTC_10x6,
         Format.ASTC_10x8, Format.ASTC_10x10, Format.ASTC_12x10, Format.ASTC_12x12]
for _i, _f in enumerate(_ASTC):
    _GL[(_f, Type.UNorm)] = (0x93B0 + _i, 0x93D0 + _i, _GL_RGBA)

# Texture::hasAlpha (Texture.cpp:467-512)
_HAS_ALPHA = {Format.R4G4B4A4, Format.B4G4R4A4, Format.R5G5B5A1, Format.B5G5R5A1, Format.A1R5G5B5,
              Format.R8G8B8A8, Format.B8G8R8A8, Format.A8B8G8R8, Format.A2R10G10B10, Format.A2B10G10R10,
              Format.R16G16B16A16, Format.R32G32B32A32, Format.BC1_RGBA, Format.BC2, Format.BC3,
              Format.BC7, Format.ETC2_R8G8B8A1, Format.ETC2_R8G8B8A8} | set(_ASTC)


def has_alpha(fmt) -> bool:
    return Format(fmt) in _HAS_ALPHA


# uncompressed formats, getFormatInfo (SaveKtx.cpp:200-630):
#   (format, type) -> (glType, glTypeSize, glFormat, (internal linear, internal sRGB), baseInternal)
# R4G4, A4R4G4B4 and B8G8R8 have no KTX form (SaveKtx.cpp:1174-1176).
_GL_UB, _GL_B, _GL_US, _GL_S, _GL_UINT, _GL_INT, _GL_HF, _GL_FL = 0x1401, 0x1400, 0x1403, 0x1402, 0x1405, 0x1404, 0x140B, 0x1406
_GL_LUM, _GL_LUMA, _GL_BGRA, _GL_RGBA_INT, _GL_BGRA_INT = 0x1909, 0x190A, 0x80E1, 0x8D99, 0x8D9B
_GLU = {
    (Format.R4G4B4A4, _U): (0x8033, 2, _GL_RGBA, (0x8056, None), _GL_RGBA),
    (Format.B4G4R4A4, _U): (0x8033, 2, _GL_BGRA, (0x8056, None), _GL_BGRA),
    (Format.R5G6B5, _U): (0x8363, 2, _GL_RGB, (0x8D62, None), _GL_RGB),
    (Format.B5G6R5, _U): (0x8364, 2, _GL_RGB, (0x8D62, None), _GL_RGB),
    (Format.R5G5B5A1, _U): (0x8034, 2, _GL_RGBA, (0x8057, None), _GL_RGBA),
    (Format.B5G5R5A1, _U): (0x8034, 2, _GL_BGRA, (0x8057, None), _GL_BGRA),
    (Format.A1R5G5B5, _U): (0x8366, 2, _GL_BGRA, (0x8057, None), _GL_BGRA),
    (Format.R8, _U): (_GL_UB, 1, _GL_RED, (0x8229, None), _GL_LUM), (Format.R8, _S): (_GL_B, 1, _GL_RED, (0x8F94, None), _GL_LUM),
    (Format.R8, _UI): (_GL_UB, 1, _GL_RED, (0x8232, None), _GL_LUM), (Format.R8, _I): (_GL_B, 1, _GL_RED, (0x8231, None), _GL_LUM),
    (Format.R8G8, _U): (_GL_UB, 1, _GL_RG, (0x822B, None), _GL_LUMA), (Format.R8G8, _S): (_GL_UB, 1, _GL_RG, (0x8F95, None), _GL_LUMA),
    (Format.R8G8, _UI): (_GL_UB, 1, _GL_RG, (0x8238, None), _GL_LUMA), (Format.R8G8, _I): (_GL_UB, 1, _GL_RG, (0x8237, None), _GL_LUMA),
    (Format.R8G8B8, _U): (_GL_UB, 1, _GL_RGB, (0x8051, 0x8C41), _GL_RGB), (Format.R8G8B8, _S): (_GL_B, 1, _GL_RGB, (0x8F96, None), _GL_RGB),
    (Format.R8G8B8, _UI): (_GL_UB, 1, _GL_RGB, (0x8D7D, None), _GL_RGB), (Format.R8G8B8, _I): (_GL_B, 1, _GL_RGB, (0x8D8F, None), _GL_RGB),
    (Format.R8G8B8A8, _U): (_GL_UB, 1, _GL_RGBA, (0x8058, 0x8C43), _GL_RGBA), (Format.R8G8B8A8, _S): (_GL_B, 1, _GL_RGBA, (0x8F97, None), _GL_RGBA),
    (Format.R8G8B8A8, _UI): (_GL_UB, 1, _GL_RGBA_INT, (0x8D7C, None), _GL_RGBA), (Format.R8G8B8A8, _I): (_GL_B, 1, _GL_RGBA_INT, (0x8D8E, None), _GL_RGBA),
    (Format.B8G8R8A8, _U): (0x8035, 4, _GL_BGRA, (0x8058, 0x8C43), _GL_BGRA),
    (Format.A8B8G8R8, _U): (0x8367, 4, _GL_RGBA, (0x8058, 0x8C43), _GL_RGBA),
    (Format.A2R10G10B10, _U): (0x8368, 4, _GL_BGRA, (0x8059, None), _GL_BGRA), (Format.A2R10G10B10, _UI): (0x8368, 4, _GL_BGRA_INT, (0x906F, None), _GL_BGRA),
    (Format.A2B10G10R10, _U): (0x8368, 4, _GL_RGBA, (0x8059, None), _GL_RGBA), (Format.A2B10G10R10, _UI): (0x8368, 4, _GL_RGBA_INT, (0x906F, None), _GL_RGBA),
    (Format.B10G11R11_UFloat, _UF): (0x8C3B, 4, _GL_RGB, (0x8C3A, None), _GL_RGB),
    (Format.E5B9G9R9_UFloat, _UF): (0x8C3E, 4, _GL_RGB, (0x8C3D, None), _GL_RGB),
}
for _f, _glf, _base, _i16, _i32 in (
        (Format.R16, _GL_RED, _GL_LUM, (0x822A, 0x8F98, 0x8234, 0x8233, 0x822D), None),
        (Format.R16G16, _GL_RG, _GL_LUMA, (0x822C, 0x8F99, 0x823A, 0x8239, 0x822F), None),
        (Format.R16G16B16, _GL_RGB, _GL_RGB, (0x8054, 0x8F9A, 0x8D77, 0x8D89, 0x881B), None),
        (Format.R16G16B16A16, _GL_RGBA, _GL_RGBA, (0x805B, 0x8F9B, 0x8D76, 0x8D88, 0x881A), None),
        (Format.R32, _GL_RED, _GL_LUM, None, (0x8236, 0x8235, 0x822E)),
        (Format.R32G32, _GL_RG, _GL_LUMA, None, (0x823C, 0x823B, 0x8230)),
        (Format.R32G32B32, _GL_RGB, _GL_RGB, None, (0x8D71, 0x8D83, 0x8815)),
        (Format.R32G32B32A32, _GL_RGBA, _GL_RGBA, None, (0x8D70, 0x8D82, 0x8814))):
    if _i16:
        for _t, _gt, _int in zip((_U, _S, _UI, _I, _F), (_GL_US, _GL_S, _GL_US, _GL_S, _GL_HF), _i16):
            _GLU[(_f, _t)] = (_gt, 2, _glf, (_int, None), _base)
    else:
        for _t, _gt, _int in zip((_UI, _I, _F), (_GL_UINT, _GL_INT, _GL_FL), _i32):
            _GLU[(_f, _t)] = (_gt, 4, _glf, (_int, None), _base)
for _f in _ASTC:                                        # the HDR profile shares the enums (SaveKtx.cpp)
    _GL[(_f, Type.UFloat)] = _GL[(_f, Type.UNorm)]

KTX_IDENTIFIER = bytes([0xAB, 0x4B, 0x54, 0x58, 0x20, 0x31, 0x31, 0xBB, 0x0D, 0x0A, 0x1A, 0x0A])
KTX_ENDIANNESS = 0x04030201


def mip_dims(width: int, height: int, levels: int):
    """(w, h) of every level: max(1, dim >> level) (Texture::width(mip), Texture.cpp:529-560)."""
    return [(max(1, width >> l), max(1, height >> l)) for l in range(levels)]


def _check_levels(fmt, typ, width, height, elements):
    """elements: list (array elements) of lists (mip levels) of bytes."""
    levels = len(elements[0])
    dims = mip_dims(width, height, levels)
    for mips in elements:
        if len(mips) != levels:
            raise ValueError("every array element needs the same number of mip levels")
        for (w, h), data in zip(dims, mips):
            if len(data) != payload_size(fmt, typ, w, h):
                raise ValueError("level %dx%d: %d bytes, expected %d" %
                                 (w, h, len(data), payload_size(fmt, typ, w, h)))
    return levels


def _as_elements(levels_or_elements):
    first = levels_or_elements[0]
    if isinstance(first, (bytes, bytearray, memoryview)) or hasattr(first, "tobytes"):
        return [list(levels_or_elements)]
    return [list(e) for e in levels_or_elements]


def _b(x) -> bytes:
    return x.tobytes() if hasattr(x, "tobytes") else bytes(x)


def write_dds(stream, fmt, typ, width: int, height: int, levels_or_elements: Sequence,
              color_space=ColorSpace.Linear, alpha=Alpha.Standard) -> int:
    """Write a DX10-header DDS.  levels_or_elements: [level0, level1, ...] payloads of one 2-D
    texture, or a list of such lists for a 2-D array.  Returns the number of bytes written."""
    fmt, typ = Format(fmt), Type(typ)
    key = (fmt, typ)
    if key not in _DXGI:
        raise ValueError("no DDS format for %s/%s (saveDds returns Unsupported)" % (fmt.name, typ.name))
    lin, srgb = _DXGI[key]
    dxgi = srgb if (ColorSpace(color_space) == ColorSpace.sRGB and srgb) else lin
    elements = [[_b(m) for m in e] for e in _as_elements(levels_or_elements)]
    levels = _check_levels(fmt, typ, width, height, elements)
    bw, _bh, bs = query(fmt, typ)
    pitch = (width + bw - 1) // bw * bs                      # computePitch, SaveDds.cpp:553-558
    flags = _DDSD_CAPS | _DDSD_HEIGHT | _DDSD_WIDTH | _DDSD_PIXELFORMAT | _DDSD_MIPMAPCOUNT | _DDSD_PITCH
    caps = _DDSCAPS_TEXTURE
    if levels > 1:
        caps |= _DDSCAPS_MIPMAP
    if levels > 1 or len(elements) > 1:
        caps |= _DDSCAPS_COMPLEX
    misc2 = _ALPHA_MODE[Alpha(alpha)] if has_alpha(fmt) else 3
    out = struct.pack("<I", DDS_MAGIC)
    out += struct.pack("<7I44x", 124, flags, height, width, pitch, 0, levels)
    out += struct.pack("<2I4s5I", 32, _DDPF_FOURCC, b"DX10", 0, 0, 0, 0, 0)
    out += struct.pack("<5I", caps, 0, 0, 0, 0)
    # arraySize = texture.depth() (SaveDds.cpp:637): 0 for a texture that is not an array
    first = levels_or_elements[0]
    is_array = not (isinstance(first, (bytes, bytearray, memoryview)) or hasattr(first, "tobytes"))
    out += struct.pack("<5I", dxgi, _DIM_TEXTURE2D, 0, len(elements) if is_array else 0, misc2)
    assert len(out) == 4 + 124 + 20
    for e in elements:          # element -> (face) -> mip, SaveDds.cpp:657-680
        for m in e:
            out += m
    stream.write(out)
    return len(out)


def read_dds(data: bytes):
    """Parse a DDS written by write_dds: dict(width, height, levels, elements, dxgi, payload offset)."""
    magic, size, flags, height, width, pitch, depth, levels = struct.unpack_from("<8I", data, 0)
    if magic != DDS_MAGIC or size != 124:
        raise ValueError("not a DDS file")
    fourcc = data[4 + 80:4 + 84]
    if fourcc != b"DX10":
        raise ValueError("only DX10-header DDS files are supported")
    dxgi, dim, misc, array, misc2 = struct.unpack_from("<5I", data, 4 + 124)
    return {"width": width, "height": height, "levels": levels, "elements": array, "dxgi": dxgi,
            "pitch": pitch, "alpha_mode": misc2, "offset": 4 + 124 + 20}


def write_ktx(stream, fmt, typ, width: int, height: int, levels_or_elements: Sequence,
              color_space=ColorSpace.Linear) -> int:
    """Write a KTX 1.1 file (block-compressed formats, and the uncompressed ones with padded rows)."""
    fmt, typ = Format(fmt), Type(typ)
    key = (fmt, typ)
    srgb_wanted = ColorSpace(color_space) == ColorSpace.sRGB
    if key in _GL:
        lin, srgb, base = _GL[key]
        gl_type, type_size, gl_format, compressed = 0, 1, 0, True
    elif key in _GLU:
        gl_type, type_size, gl_format, (lin, srgb), base = _GLU[key]
        compressed = False
    else:
        raise ValueError("no KTX format for %s/%s (saveKtx returns Unsupported)" % (fmt.name, typ.name))
    internal = srgb if (srgb_wanted and srgb) else lin
    elements = [[_b(m) for m in e] for e in _as_elements(levels_or_elements)]
    levels = _check_levels(fmt, typ, width, height, elements)
    is_array = len(elements) > 1
    out = KTX_IDENTIFIER + struct.pack("<I", KTX_ENDIANNESS)
    out += struct.pack("<5I", gl_type, type_size, gl_format, internal, base)
    out += struct.pack("<7I", width, height, 0, len(elements) if is_array else 0, 1, levels, 0)
    bpp = query(fmt, typ)[2]
    dims = mip_dims(width, height, levels)
    for l in range(levels):
        w, h = dims[l]
        if compressed:
            size = sum(len(e[l]) for e in elements)          # SaveKtx.cpp:1224-1248
            pad = 0
        else:
            row = w*bpp                                      # scanlines are 4-byte aligned (:1240-1285)
            pad = (4 - row % 4) % 4
            size = (row + pad)*h*len(elements)
        assert size % 4 == 0
        out += struct.pack("<I", size)
        for e in elements:                                   # depth (array element) -> face
            if pad == 0:
                out += e[l]
            else:
                for y in range(h):
                    out += e[l][y*row:(y + 1)*row] + b"\0"*pad
    stream.write(out)
    return len(out)


# ---- PVR v3 (savePvr, SavePvr.cpp:478-600) ---------------------------------------------------

def _fourcc(a, b, c, d) -> int:
    o = lambda x: x if isinstance(x, int) else ord(x)
    return o(a) | (o(b) << 8) | (o(c) << 16) | (o(d) << 24)


def _pvr_generic(*pairs) -> int:
    """PVR_GENERIC_FORMAT (SavePvr.cpp:23-27): channel letters in the low dword, bit counts above."""
    v = 0
    for i, (ch, bits) in enumerate(pairs):
        v |= (ord(ch) if ch else 0) << (8*i)
        v |= bits << (32 + 8*i)
    return v


# PvrSpecialFormat (SavePvr.cpp:53-105)
_PVR_SPECIAL = {
    Format.ETC1: 6, Format.BC1_RGB: 7, Format.BC1_RGBA: 7, Format.BC2: 9, Format.BC3: 11, Format.BC4: 12,
    Format.BC5: 13, Format.BC6H: 14, Format.BC7: 15, Format.E5B9G9R9_UFloat: 19,
    Format.ETC2_R8G8B8: 22, Format.ETC2_R8G8B8A8: 23, Format.ETC2_R8G8B8A1: 24, Format.EAC_R11: 25,
    Format.EAC_R11G11: 26,
}
for _i, _f in enumerate(_ASTC):
    _PVR_SPECIAL[_f] = 27 + _i
_PVR_GENERIC = {
    Format.R4G4: "r4g4", Format.R4G4B4A4: "r4g4b4a4", Format.B4G4R4A4: "b4g4r4a4", Format.A4R4G4B4: "a4r4g4b4",
    Format.R5G6B5: "r5g6b5", Format.B5G6R5: "b5g6r5", Format.R5G5B5A1: "r5g5b5a1", Format.B5G5R5A1: "b5g5r5a1",
    Format.A1R5G5B5: "a1r5g5b5", Format.R8: "r8", Format.R8G8: "r8g8", Format.R8G8B8: "r8g8b8",
    Format.B8G8R8: "b8g8r8", Format.R8G8B8A8: "r8g8b8a8", Format.B8G8R8A8: "b8g8r8a8",
    Format.A8B8G8R8: "a8b8g8r8", Format.A2R10G10B10: "a2r10g10b10", Format.A2B10G10R10: "a2b10g10r10",
    Format.R16: "r16", Format.R16G16: "r16g16", Format.R16G16B16: "r16g16b16",
    Format.R16G16B16A16: "r16g16b16a16", Format.R32: "r32", Format.R32G32: "r32g32",
    Format.R32G32B32: "r32g32b32", Format.R32G32B32A32: "r32g32b32a32", Format.B10G11R11_UFloat: "b10g11r11",
}


def pvr_pixel_format(fmt, alpha=Alpha.Standard) -> int:
    """getPixelFormat (SavePvr.cpp:272-470)."""
    fmt = Format(fmt)
    if fmt in _PVR_GENERIC:
        import re
        return _pvr_generic(*[(c, int(b)) for c, b in re.findall(r"([rgba])(\d+)", _PVR_GENERIC[fmt])])
    if Alpha(alpha) == Alpha.PreMultiplied and fmt in (Format.BC2, Format.BC3):
        return 8 if fmt == Format.BC2 else 10                # DXT2 / DXT4
    return _PVR_SPECIAL[fmt]


_PVR_BYTE = {Format.R4G4, Format.R8, Format.R8G8, Format.R8G8B8, Format.B8G8R8, Format.R8G8B8A8,
             Format.B8G8R8A8, Format.A8B8G8R8}
_PVR_SHORT = {Format.R4G4B4A4, Format.B4G4R4A4, Format.A4R4G4B4, Format.R5G6B5, Format.B5G6R5,
              Format.R5G5B5A1, Format.B5G5R5A1, Format.A1R5G5B5, Format.R16, Format.R16G16,
              Format.R16G16B16, Format.R16G16B16A16}
_PVR_INT = {Format.A2R10G10B10, Format.A2B10G10R10, Format.R32, Format.R32G32, Format.R32G32B32,
            Format.R32G32B32A32}


def pvr_channel_type(fmt, typ) -> int:
    """getChannelType (SavePvr.cpp:107-268): PvrChannelType index."""
    fmt, typ = Format(fmt), Type(typ)
    if typ == Type.UFloat:
        return 13
    if typ == Type.Float:
        return 12
    norm = typ in (Type.UNorm, Type.SNorm)
    signed = typ in (Type.SNorm, Type.Int)
    byte_like = fmt in _PVR_BYTE or (norm and fmt in (Format.BC4, Format.BC5))
    short_like = fmt in _PVR_SHORT or (norm and fmt in (Format.EAC_R11, Format.EAC_R11G11))
    if byte_like:
        base = 0
    elif short_like:
        base = 4
    elif fmt in _PVR_INT:
        base = 8
    else:                                                     # the default: rows of the switch
        return {Type.UNorm: 0, Type.SNorm: 1, Type.UInt: 2, Type.Int: 2}[typ]
    return base + (0 if norm else 2) + (1 if signed else 0)


def write_pvr(stream, fmt, typ, width: int, height: int, levels_or_elements: Sequence,
              color_space=ColorSpace.Linear, alpha=Alpha.Standard) -> int:
    """Write a PVR v3 file of a 2-D texture or 2-D array (surface order mip -> element)."""
    fmt, typ = Format(fmt), Type(typ)
    elements = [[_b(m) for m in e] for e in _as_elements(levels_or_elements)]
    levels = _check_levels(fmt, typ, width, height, elements)
    is_array = len(elements) > 1
    out = struct.pack("<II", _fourcc("P", "V", "R", 3), 0x2 if Alpha(alpha) == Alpha.PreMultiplied else 0)
    out += struct.pack("<Q", pvr_pixel_format(fmt, alpha))
    out += struct.pack("<II", 1 if ColorSpace(color_space) == ColorSpace.sRGB else 0, pvr_channel_type(fmt, typ))
    out += struct.pack("<6I", height, width, 1, len(elements) if is_array else 1, 1, levels)
    meta = b""
    if fmt in (Format.BC1_RGB, Format.BC1_RGBA):              # BC1 alpha is told apart by metadata
        code = _fourcc("B", "C", "1", "A") if fmt == Format.BC1_RGBA else _fourcc("B", "C", "1", 0)
        meta += struct.pack("<4I", _fourcc("C", "T", "F", "S"), code, 4, 0)
    if is_array:
        meta += struct.pack("<4I", _fourcc("C", "T", "F", "S"), _fourcc("A", "R", "R", "Y"), 4, 0)
    out += struct.pack("<I", len(meta)) + meta
    for l in range(levels):
        for e in elements:
            out += e[l]
    stream.write(out)
    return len(out)


def read_pvr(data: bytes):
    """Parse the header of a PVR v3 file (round-trip tests)."""
    version, flags, pixfmt, cspace, chtype, height, width, depth, surfaces, faces, levels, meta = \
        struct.unpack_from("<IIQIIIIIIIII", data, 0)
    if version != _fourcc("P", "V", "R", 3):
        raise ValueError("not a PVR v3 file")
    return {"flags": flags, "pixel_format": pixfmt, "color_space": cspace, "channel_type": chtype,
            "width": width, "height": height, "depth": depth, "elements": surfaces, "faces": faces,
            "levels": levels, "metadata": data[52:52 + meta], "offset": 52 + meta}


# ---- general [level][depth][face] writers: cube maps, cube arrays, 3-D textures -------------
# (Texture::data(face, level, depth); depth(level) = max(1, depth >> level) for Dim3D, the array
#  size otherwise, Texture.cpp:1207-1216)

_DIM_TEXTURE3D = 4
_DIM_TEXTURE1D = 2
_DDSD_DEPTH = 0x800000
_DDSCAPS2_CUBE_ALL = 0x200 | 0x400 | 0x800 | 0x1000 | 0x2000 | 0x4000 | 0x8000
_DDSCAPS2_VOLUME = 0x200000
_DDS_MISC_CUBEMAP = 0x4


class TextureLayout:
    """Shape of a texture as the reference's writers see it."""

    def __init__(self, fmt, typ, width, height, surfaces, dimension="2d", depth=0, is_array=None):
        self.fmt, self.typ = Format(fmt), Type(typ)
        self.width, self.height = int(width), int(height)
        self.dimension = dimension
        if dimension not in ("1d", "2d", "cube", "3d"):
            raise ValueError("dimension must be '1d', '2d', 'cube' or '3d'")
        self.faces = 6 if dimension == "cube" else 1
        self.depth = int(depth)                                  # Texture::depth(): 0 = not an array / not 3-D
        self.is_array = (dimension != "3d" and self.depth > 0) if is_array is None else bool(is_array)
        if dimension == "3d" and self.depth < 1:
            raise ValueError("a 3-D texture needs a depth")
        self.surfaces = [[[_b(f) for f in d] for d in lvl] for lvl in surfaces]
        self.levels = len(self.surfaces)
        for l, lvl in enumerate(self.surfaces):
            w, h = max(1, self.width >> l), max(1, self.height >> l)
            if len(lvl) != self.depth_at(l):
                raise ValueError("level %d: %d depth entries, expected %d" % (l, len(lvl), self.depth_at(l)))
            for d in lvl:
                if len(d) != self.faces:
                    raise ValueError("level %d: %d faces, expected %d" % (l, len(d), self.faces))
                for f in d:
                    if len(f) != payload_size(self.fmt, self.typ, w, h):
                        raise ValueError("level %d: %d bytes, expected %d" % (l, len(f), payload_size(self.fmt, self.typ, w, h)))

    def depth_at(self, level):
        if self.dimension == "3d":
            return max(self.depth >> level, 1)
        return max(self.depth, 1)


def write_dds_texture(stream, tex: TextureLayout, color_space=ColorSpace.Linear, alpha=Alpha.Standard) -> int:
    """saveDds (SaveDds.cpp:565-683) for any dimension: surface order element -> face -> level ->
    volume slice (:657-680)."""
    key = (tex.fmt, tex.typ)
    if key not in _DXGI:
        raise ValueError("no DDS format for %s/%s (saveDds returns Unsupported)" % (tex.fmt.name, tex.typ.name))
    lin, srgb = _DXGI[key]
    dxgi = srgb if (ColorSpace(color_space) == ColorSpace.sRGB and srgb) else lin
    bw, _bh, bs = query(tex.fmt, tex.typ)
    pitch = (tex.width + bw - 1) // bw * bs
    is3d = tex.dimension == "3d"
    flags = _DDSD_CAPS | _DDSD_HEIGHT | _DDSD_WIDTH | _DDSD_PIXELFORMAT | _DDSD_MIPMAPCOUNT | _DDSD_PITCH
    if is3d:
        flags |= _DDSD_DEPTH                                            # :579-580
    caps = _DDSCAPS_TEXTURE
    if tex.levels > 1:
        caps |= _DDSCAPS_MIPMAP
    if tex.levels > 1 or is3d or tex.is_array:                          # :592-596 (a lone cube map is not "complex" there)
        caps |= _DDSCAPS_COMPLEX
    caps2 = _DDSCAPS2_CUBE_ALL if tex.dimension == "cube" else (_DDSCAPS2_VOLUME if is3d else 0)
    misc2 = _ALPHA_MODE[Alpha(alpha)] if has_alpha(tex.fmt) else 3
    out = struct.pack("<I", DDS_MAGIC)
    out += struct.pack("<7I44x", 124, flags, tex.height, tex.width, pitch, tex.depth if is3d else 0, tex.levels)
    out += struct.pack("<2I4s5I", 32, _DDPF_FOURCC, b"DX10", 0, 0, 0, 0, 0)
    out += struct.pack("<5I", caps, caps2, 0, 0, 0)
    out += struct.pack("<5I", dxgi, _DIM_TEXTURE3D if is3d else (_DIM_TEXTURE1D if tex.dimension == "1d" else _DIM_TEXTURE2D),
                       _DDS_MISC_CUBEMAP if tex.dimension == "cube" else 0,
                       1 if is3d else tex.depth, misc2)   # arraySize = texture.depth(), 0 for a non-array (:637)
    elements = max(tex.depth, 1) if tex.is_array else 1
    for element in range(elements):
        for face in range(tex.faces):
            for level in range(tex.levels):
                volumes = tex.depth_at(level) if is3d else 1
                for volume in range(volumes):
                    out += tex.surfaces[level][volume + element][face]
    stream.write(out)
    return len(out)


def write_ktx_texture(stream, tex: TextureLayout, color_space=ColorSpace.Linear) -> int:
    """saveKtx (SaveKtx.cpp:1189-1290) for any dimension: per level imageSize, then depth -> face;
    imageSize counts ONE face of a non-array cube map (:1224-1240, the KTX rule)."""
    key = (tex.fmt, tex.typ)
    srgb_wanted = ColorSpace(color_space) == ColorSpace.sRGB
    if key in _GL:
        lin, srgb, base = _GL[key]
        gl_type, type_size, gl_format, compressed = 0, 1, 0, True
    elif key in _GLU:
        gl_type, type_size, gl_format, (lin, srgb), base = _GLU[key]
        compressed = False
    else:
        raise ValueError("no KTX format for %s/%s (saveKtx returns Unsupported)" % (tex.fmt.name, tex.typ.name))
    internal = srgb if (srgb_wanted and srgb) else lin
    out = KTX_IDENTIFIER + struct.pack("<I", KTX_ENDIANNESS)
    out += struct.pack("<5I", gl_type, type_size, gl_format, internal, base)
    out += struct.pack("<7I", tex.width, 0 if tex.dimension == "1d" else tex.height,      # :1209
                       tex.depth if tex.dimension == "3d" else 0,
                       tex.depth if tex.is_array else 0, tex.faces, tex.levels, 0)
    bpp = query(tex.fmt, tex.typ)[2]
    for l in range(tex.levels):
        w, h = max(1, tex.width >> l), max(1, tex.height >> l)
        row = w*bpp
        pad = 0 if compressed else (4 - row % 4) % 4
        one = len(tex.surfaces[l][0][0]) if compressed else (row + pad)*h
        size = one*tex.depth_at(l)
        if tex.is_array:
            size *= tex.faces
        assert size % 4 == 0
        out += struct.pack("<I", size)
        for d in range(tex.depth_at(l)):
            for f in range(tex.faces):
                data = tex.surfaces[l][d][f]
                if pad == 0:
                    out += data
                else:
                    for y in range(h):
                        out += data[y*row:(y + 1)*row] + b"\0"*pad
    stream.write(out)
    return len(out)


def write_pvr_texture(stream, tex: TextureLayout, color_space=ColorSpace.Linear, alpha=Alpha.Standard) -> int:
    """savePvr (SavePvr.cpp:478-600) for any dimension: surface order level -> depth -> face."""
    out = struct.pack("<II", _fourcc("P", "V", "R", 3), 0x2 if Alpha(alpha) == Alpha.PreMultiplied else 0)
    out += struct.pack("<Q", pvr_pixel_format(tex.fmt, alpha))
    out += struct.pack("<II", 1 if ColorSpace(color_space) == ColorSpace.sRGB else 0, pvr_channel_type(tex.fmt, tex.typ))
    out += struct.pack("<6I", tex.height, tex.width, tex.depth if tex.dimension == "3d" else 1,
                       tex.depth if tex.is_array else 1, tex.faces, tex.levels)
    meta = b""
    if tex.fmt in (Format.BC1_RGB, Format.BC1_RGBA):
        code = _fourcc("B", "C", "1", "A") if tex.fmt == Format.BC1_RGBA else _fourcc("B", "C", "1", 0)
        meta += struct.pack("<4I", _fourcc("C", "T", "F", "S"), code, 4, 0)
    if tex.is_array:
        meta += struct.pack("<4I", _fourcc("C", "T", "F", "S"), _fourcc("A", "R", "R", "Y"), 4, 0)
    if tex.dimension == "1d":                                          # SavePvr.cpp:568-575
        meta += struct.pack("<4I", _fourcc("C", "T", "F", "S"), _fourcc("D", "I", "M", "1"), 4, 0)
    out += struct.pack("<I", len(meta)) + meta
    for l in range(tex.levels):
        for d in range(tex.depth_at(l)):
            for f in range(tex.faces):
                out += tex.surfaces[l][d][f]
    stream.write(out)
    return len(out)
