"""The reference's own TextureTest cases that need no image arithmetic (lib/test/TextureTest.cpp:
420-472 Create / SetImages / SetImagesCube) and the static helpers of Texture, run through the
mirror (cuttlefish_amd/texture.py).  Image(format, w, h) of the reference is an (h, w, 4) array."""
import json
import os

import numpy as np

from cuttlefish_amd import ColorSpace, Format, Type
from cuttlefish_amd.texture import (CubeFace, Dimension, FileType, SaveResult, Texture, image_index)


def image(w, h, dtype=np.float32):
    return np.zeros((h, w, 4), dtype)


def test_create():
    t = Texture(Dimension.Dim2D, 10, 15, 0)
    assert t.dimension() == Dimension.Dim2D
    assert (t.width(), t.height(), t.depth()) == (10, 15, 1)
    assert not t.is_array()
    assert t.mip_level_count() == 1 and t.face_count() == 1
    assert t.initialize(Dimension.Cube, 15, 10, 5, Texture.allMipLevels)
    assert t.dimension() == Dimension.Cube
    assert (t.width(), t.height(), t.depth()) == (15, 10, 5)
    assert t.is_array()
    assert t.mip_level_count() == 4 and t.face_count() == 6


def test_invalid_textures():
    t = Texture()
    assert not t.is_valid() and not t
    assert (t.width(), t.height(), t.depth(), t.mip_level_count(), t.face_count()) == (0, 0, 0, 0, 0)
    assert not t.images_complete() and not t.set_image(image(4, 4))
    assert not t.initialize(Dimension.Dim2D, 0, 4) and not t.initialize(Dimension.Dim3D, 4, 4, 0)
    assert t.initialize(Dimension.Dim3D, 4, 4, 2) and t.is_valid() and not t.is_array()
    assert t.save("x.dds") == SaveResult.Invalid                       # not converted
    assert t.width(1) == 0 and t.get_image(0, 5) is None


def test_set_images():
    t = Texture(Dimension.Dim2D, 15, 10, 5)
    assert not t.set_image(image(10, 15))
    for i in range(5):
        assert not t.images_complete()
        assert t.set_image(image(15, 10), 0, i)
    assert t.images_complete()
    assert not t.set_image(image(15, 10), 0, 5) and not t.set_image(image(15, 10), 1, 0)
    assert not t.set_image(image(15, 10), CubeFace.NegX, 0, 0)          # not a cube map
    assert t.set_image(image(15, 10), CubeFace.PosX, 0, 0)              # PosX addresses the only face


def test_set_images_cube():
    t = Texture(Dimension.Cube, 15, 10, 5)
    assert not t.set_image(image(10, 15), CubeFace.PosX)
    assert not t.set_image(image(15, 10))                               # a cube map needs the face
    for f in CubeFace:
        for j in range(5):
            assert not t.images_complete()
            assert t.set_image(image(15, 10), f, 0, j)
    assert t.images_complete()
    assert t.get_image(CubeFace.NegZ, 0, 4) is not None and t.get_image(0, 0) is None


def test_set_images_keep_the_colour_space_and_become_rgbaf_sources():
    t = Texture(Dimension.Dim2D, 15, 10, 5, 1, ColorSpace.sRGB)
    assert t.color_space() == ColorSpace.sRGB
    for i in range(5):
        assert t.set_image(image(15, 10, np.uint8), 0, i)
        assert t.get_image(0, i).shape == (10, 15, 4)
    assert t.set_image(image(15, 10, np.float64), 0, 0) and t.get_image(0, 0).dtype == np.float32
    assert t.images_complete()


def test_mip_geometry_of_every_dimension():
    t = Texture(Dimension.Dim3D, 15, 10, 5, Texture.allMipLevels)
    assert t.mip_level_count() == 4
    assert [(t.width(m), t.height(m), t.depth(m)) for m in range(5)] == \
        [(15, 10, 5), (7, 5, 2), (3, 2, 1), (1, 1, 1), (0, 0, 0)]
    assert Texture.max_mipmap_levels(Dimension.Dim3D, 4, 4, 64) == 7
    assert Texture.max_mipmap_levels(Dimension.Dim2D, 4, 4, 64) == 3
    a = Texture(Dimension.Dim2D, 16, 8, 3, Texture.allMipLevels)
    assert a.mip_level_count() == 5 and [a.depth(m) for m in range(5)] == [3]*5
    one = Texture(Dimension.Dim1D, 33, 1, 0, 3)
    assert one.mip_level_count() == 3 and [one.width(m) for m in range(3)] == [33, 16, 8]


def test_statics():
    assert Texture.file_type("a/b/tex.DDS") == FileType.DDS and Texture.file_type("x.ktx") == FileType.KTX
    assert Texture.file_type("x.Pvr") == FileType.PVR and Texture.file_type("x.png") == FileType.Auto
    assert Texture.file_type("dds") == FileType.Auto
    assert Texture.block_width(Format.ASTC_10x6) == 10 and Texture.block_height(Format.ASTC_10x6) == 6
    assert Texture.block_size(Format.BC1_RGB) == 8 and Texture.block_size(Format.R32G32B32A32) == 16
    assert Texture.min_width(Format.BC7) == 4 and Texture.min_height(Format.ASTC_8x5) == 5
    assert Texture.min_width(Format.R8) == 1
    assert Texture.has_alpha(Format.BC3) and not Texture.has_alpha(Format.BC4)
    assert Texture.has_native_srgb(Format.BC7, Type.UNorm) and not Texture.has_native_srgb(Format.BC6H, Type.UFloat)
    assert Texture.is_format_valid(Format.BC6H, Type.UFloat) and not Texture.is_format_valid(Format.BC6H, Type.UNorm)
    assert image_index(2, 3) == (0, 2, 3) and image_index(CubeFace.NegY, 1) == (3, 1, 0)


def test_is_format_valid_per_file_type_matches_the_reference_save_tables():
    """Texture::isFormatValid(format, type, fileType) against the outcome tables of
    lib/test/TextureSaveTest.cpp (fixture tests/golden/save_expectations.json)."""
    exp = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "save_expectations.json")))
    n = 0
    for kind, table in exp.items():
        for key, ok in table.items():
            fname, tname = key.split("/")
            if fname.startswith("PVRTC"):
                continue
            assert Texture.is_format_valid(getattr(Format, fname), getattr(Type, tname),
                                           getattr(FileType, kind)) == ok, (kind, key)
            n += 1
    assert n > 250
    assert not Texture.is_format_valid(Format.BC7, Type.UNorm, FileType.Auto)


def _test_colour(w, h):
    """getTestColor of the reference's tests (lib/test/TextureTest.cpp:53-61)"""
    x, y = np.meshgrid(np.arange(w), np.arange(h))
    return np.stack([x/(w - 1), y/(h - 1), (w - 1 - x)/(w - 1), (h - 1 - y)/(h - 1)], axis=-1)


def test_adjust_image_value_range():
    """TextureTest.AdjustImageValueRangeUNorm / UInt / Int (lib/test/TextureTest.cpp:63-418) for the
    RGBA images this path carries: SNorm remaps an integer-origin image to [-1, 1], UInt scales to the
    original range, Int also offsets; float-origin images and UNorm are untouched."""
    from cuttlefish_amd.texture import ImageFormat

    def std_round(v):                                            # half away from zero, like std::round
        return np.sign(v)*np.floor(np.abs(v) + np.float32(0.5))
    ref = _test_colour(14, 15)
    img = ref.astype(np.float32)
    same = Texture.adjust_image_value_range(img, Type.UNorm, ImageFormat.RGBA8)
    assert np.array_equal(same, img)
    sn = Texture.adjust_image_value_range(img, Type.SNorm, ImageFormat.RGBA8)
    assert np.allclose(sn, ref*2.0 - 1.0, atol=1e-6)
    assert np.array_equal(Texture.adjust_image_value_range(img, Type.SNorm, ImageFormat.RGBAF), img)   # float origin
    for fmt, maxv, gmax, bits, gbits, ch in ((ImageFormat.RGBA8, 255, 255, 8, 8, 4), (ImageFormat.RGBA16, 65535, 65535, 16, 16, 4),
                                            (ImageFormat.RGB565, 31, 63, 5, 6, 3), (ImageFormat.RGB5, 31, 31, 5, 5, 3)):
        src = img[..., :ch]
        ui = Texture.adjust_image_value_range(src, Type.UInt, fmt)
        ii = Texture.adjust_image_value_range(src, Type.Int, fmt)
        for c in range(ch):
            m = gmax if c == 1 else maxv
            b = gbits if c == 1 else bits
            want = std_round(src[..., c].astype(np.float32)*np.float32(m))
            assert np.array_equal(ui[..., c], want), (fmt, c)
            assert np.array_equal(ii[..., c], std_round(src[..., c].astype(np.float32)*np.float32(m) + np.float32(-(1 << (b - 1))))), (fmt, c)
    u8 = (ref*255 + 0.5).astype(np.uint8)
    a = Texture.adjust_image_value_range(u8, Type.Int)                   # names its own format
    assert np.array_equal(a, std_round((u8.astype(np.float64)/255.0).astype(np.float32)*np.float32(255.0) - np.float32(128.0)))
    assert a.min() == -128.0 and a.max() == 127.0
