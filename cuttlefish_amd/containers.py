"""DDS (DX10 header) and KTX 1.1 writers for block-compressed payloads -- SURVEY section 8(f)
row 2: "container writers ... so outputs open in standard viewers".

Pure serialisation, host side, mirrors the reference writers field for field:
  saveDds   lib/src/SaveDds.cpp:565-683   (header flags :578-598, DX10 header :604-652,
                                            surface order element -> face -> mip :657-680)
  saveKtx   lib/src/SaveKtx.cpp:1189-1290 (header :1198-1221, per level imageSize :1224-1248,
                                            then depth -> face payloads :1250-1262)
Only 2-D textures and 2-D arrays of the formats this backend encodes are covered (cube maps
and 3-D textures are containers of the same surfaces in a different order: not needed by the
hot path's tests).  `read_dds` parses what `write_dds` emits (round-trip tests; Pillow is the
independent reader used by tests/test_containers.py).
"""
from __future__ import annotations

import struct
from typing import List, Sequence

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
_ASTC = [Format.ASTC_4x4, Format.ASTC_5x4, Format.ASTC_5x5, Format.ASTC_6x5, Format.ASTC_6x6,
         Format.ASTC_8x5, Format.ASTC_8x6, Format.ASTC_8x8, Format.ASTC_10x5, Format.ASTC_10x6,
         Format.ASTC_10x8, Format.ASTC_10x10, Format.ASTC_12x10, Format.ASTC_12x12]
for _i, _f in enumerate(_ASTC):
    _GL[(_f, Type.UNorm)] = (0x93B0 + _i, 0x93D0 + _i, _GL_RGBA)

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
    has_alpha = fmt in (Format.BC1_RGBA, Format.BC2, Format.BC3, Format.BC7)
    misc2 = _ALPHA_MODE[Alpha(alpha)] if has_alpha else 3
    out = struct.pack("<I", DDS_MAGIC)
    out += struct.pack("<7I44x", 124, flags, height, width, pitch, 0, levels)
    out += struct.pack("<2I4s5I", 32, _DDPF_FOURCC, b"DX10", 0, 0, 0, 0, 0)
    out += struct.pack("<5I", caps, 0, 0, 0, 0)
    out += struct.pack("<5I", dxgi, _DIM_TEXTURE2D, 0, len(elements), misc2)
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
    """Write a KTX 1.1 file (compressed formats only)."""
    fmt, typ = Format(fmt), Type(typ)
    key = (fmt, typ)
    if key not in _GL:
        raise ValueError("no KTX format for %s/%s (saveKtx returns Unsupported)" % (fmt.name, typ.name))
    lin, srgb, base = _GL[key]
    internal = srgb if (ColorSpace(color_space) == ColorSpace.sRGB and srgb) else lin
    elements = [[_b(m) for m in e] for e in _as_elements(levels_or_elements)]
    levels = _check_levels(fmt, typ, width, height, elements)
    is_array = len(elements) > 1
    out = KTX_IDENTIFIER + struct.pack("<I", KTX_ENDIANNESS)
    out += struct.pack("<5I", 0, 1, 0, internal, base)       # type, typeSize, format, internal, base
    out += struct.pack("<7I", width, height, 0, len(elements) if is_array else 0, 1, levels, 0)
    for l in range(levels):
        size = sum(len(e[l]) for e in elements)              # SaveKtx.cpp:1224-1248
        assert size % 4 == 0
        out += struct.pack("<I", size)
        for e in elements:                                   # depth (array element) -> face
            out += e[l]
    stream.write(out)
    return len(out)
