"""The reference's TextureTest cases that generate mips, convert and save (lib/test/TextureTest.cpp:
539-805 GenerateMipmaps / GenerateMipmapsCustomMips / Generate3DMipmaps / Generate3DMipmapsCustomMips),
run through the mirror with the GPU doing the resizing and the encoding, plus the new
cfhip_resize_device against the oracle's Image::resize and whole-texture convert + save."""
import struct

import numpy as np
import pytest

import oracle_lib as O
from cuttlefish_amd import ColorSpace, Context, Format, Type, api, synth
from cuttlefish_amd.texture import (CubeFace, CustomMipImage, Dimension, FileType, MipReplacement,
                                    SaveResult, Texture, image_index)

pytestmark = pytest.mark.gpu
BOX = api.ResizeFilter.Box


def image(w, h):
    return np.zeros((h, w, 4), np.float32)


def solid(size, r, g, b):
    im = np.zeros((size, size, 4), np.float32)
    im[...] = (r, g, b, 1.0)
    return im


def test_generate_mipmaps_cube_array():
    t = Texture(Dimension.Cube, 15, 10, 5)
    assert not t.set_image(image(10, 15))
    for f in CubeFace:
        for j in range(5):
            assert not t.images_complete()
            assert t.set_image(image(15, 10), f, 0, j)
    assert t.images_complete()
    assert t.generate_mipmaps()
    assert t.images_complete()
    assert t.mip_level_count() == 4
    for mip, (w, h) in enumerate(((15, 10), (7, 5), (3, 2), (1, 1))):
        assert t.get_image(CubeFace.PosX, mip, 1).shape == (h, w, 4)


def test_generate_mipmaps_custom_mips():
    size = 32
    t = Texture(Dimension.Dim2D, size, size)
    red, green, blue = solid(size, 1, 0, 0), solid(size, 0, 1, 0), solid(size, 0, 0, 1)
    assert t.set_image(red)
    mips = {image_index(1): CustomMipImage(green, MipReplacement.Continue),
            image_index(2): CustomMipImage(blue, MipReplacement.Once),
            image_index(3): CustomMipImage(red, MipReplacement.Once)}
    assert t.generate_mipmaps(BOX, Texture.allMipLevels, mips)
    assert t.images_complete() and t.mip_level_count() == 6
    want = {1: (0, 1, 0), 2: (0, 0, 1), 3: (1, 0, 0), 4: (0, 1, 0), 5: (0, 1, 0)}
    for mip, rgb in want.items():
        im = t.get_image(mip)
        assert im.shape == (size >> mip, size >> mip, 4)
        assert tuple(im[0, 0, :3]) == rgb, mip
    mips[image_index(1)].image = None
    assert not t.generate_mipmaps(BOX, Texture.allMipLevels, mips)


def test_generate_3d_mipmaps():
    t = Texture(Dimension.Dim3D, 15, 10, 5)
    assert not t.set_image(image(10, 15))
    for j in range(5):
        assert not t.images_complete()
        assert t.set_image(image(15, 10), 0, j)
    assert t.images_complete() and t.generate_mipmaps() and t.images_complete()
    assert t.mip_level_count() == 4
    for mip, (w, h) in enumerate(((15, 10), (7, 5), (3, 2), (1, 1))):
        assert t.get_image(mip, 0).shape == (h, w, 4)
    assert t.get_image(1, 2) is None and t.get_image(1, 1) is not None
    assert t.get_image(2, 1) is None and t.get_image(2, 0) is not None
    assert t.get_image(3, 1) is None and t.get_image(3, 0) is not None


def test_generate_3d_mipmaps_custom_mips():
    size = 32
    t = Texture(Dimension.Dim3D, size, size, size)
    red, green, blue = solid(size, 1, 0, 0), solid(size, 0, 1, 0), solid(size, 0, 0, 1)
    for d in range(size):
        assert t.set_image(red, 0, d)
    mips = {image_index(1): CustomMipImage(green, MipReplacement.Continue),
            image_index(2): CustomMipImage(blue, MipReplacement.Once),
            image_index(3): CustomMipImage(red, MipReplacement.Once)}
    assert not t.generate_mipmaps(BOX, Texture.allMipLevels, mips)        # one slice of a level only
    for d in range(1, size//2):
        mips[image_index(1, d)] = CustomMipImage(green, MipReplacement.Once)
    for d in range(1, size//4):
        mips[image_index(2, d)] = CustomMipImage(blue, MipReplacement.Once)
    for d in range(1, size//8):
        mips[image_index(3, d)] = CustomMipImage(red, MipReplacement.Once)
    assert not t.generate_mipmaps(BOX, Texture.allMipLevels, mips)        # mixed replacement modes
    for d in range(1, size//2):
        mips[image_index(1, d)].replacement = MipReplacement.Continue
    assert t.generate_mipmaps(BOX, Texture.allMipLevels, mips)
    assert t.images_complete() and t.mip_level_count() == 6
    want = {1: (0, 1, 0), 2: (0, 0, 1), 3: (1, 0, 0), 4: (0, 1, 0), 5: (0, 1, 0)}
    for mip, rgb in want.items():
        for d in range(size >> mip):
            im = t.get_image(mip, d)
            assert im.shape == (size >> mip, size >> mip, 4)
            assert tuple(im[0, 0, :3]) == rgb, (mip, d)


@pytest.mark.parametrize("filt", list(api.ResizeFilter))
def test_resize_device_matches_the_oracle(filt):
    """cfhip_resize_device == Image::resize as the oracle restates it: any size to any size, both
    colour spaces, all pixel types."""
    import torch
    src = synth.photo(37, 23, seed=3).astype(np.float32)/np.float32(255.0)
    with Context(0) as ctx:
        for cs in (ColorSpace.Linear, ColorSpace.sRGB):
            for (w, h) in ((18, 11), (9, 23), (37, 5), (1, 1), (50, 31), (37, 23)):
                d_src = torch.from_numpy(src).cuda()
                d_dst = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
                ctx.resize_device(d_src.data_ptr(), api.PixelType.RGBA32F, 37, 23, src.strides[0],
                                  d_dst.data_ptr(), w, h, color_space=cs, filter=int(filt))
                got = d_dst.cpu().numpy()
                if (w, h) == (37, 23):
                    assert np.array_equal(got, src)                       # Image.cpp:1330-1334
                    continue
                ref = O.resize_rgbaf(src, w, h, int(filt), int(cs))
                if cs == ColorSpace.Linear:
                    assert np.array_equal(got, ref), (filt, cs, w, h)
                else:                                                     # pow(): a few ulp between libm and the GPU
                    assert np.allclose(got, ref, rtol=0, atol=4e-7), (filt, cs, w, h)
        u8 = synth.photo(16, 16, seed=1)
        d_src = torch.from_numpy(u8).cuda()
        d_dst = torch.empty((16, 16, 4), dtype=torch.float32, device="cuda")
        ctx.resize_device(d_src.data_ptr(), api.PixelType.RGBA8, 16, 16, 64, d_dst.data_ptr(), 16, 16, filter=int(filt))
        assert np.array_equal(d_dst.cpu().numpy(), (u8.astype(np.float64)/255.0).astype(np.float32))


def _surfaces(t):
    return [(m, d, f) for m in range(t.mip_level_count()) for d in range(t.depth(m)) for f in range(t.face_count())]


@pytest.mark.parametrize("dim,depth", [(Dimension.Dim2D, 0), (Dimension.Dim2D, 3), (Dimension.Cube, 0),
                                       (Dimension.Cube, 2), (Dimension.Dim3D, 6), (Dimension.Dim1D, 0)])
def test_convert_and_save_every_dimension(dim, depth, tmp_path):
    """generate -> convert (one cfhip_encode for every surface) -> save: every payload equals the
    oracle's encoding of that surface, and the three containers carry them in the reference's
    surface orders (SaveDds.cpp:657-680, SaveKtx.cpp:1250-1262, SavePvr.cpp:580-595)."""
    w, h = (24, 1) if dim == Dimension.Dim1D else (24, 20)
    t = Texture(dim, w, h, depth)
    k = 0
    for d in range(t.depth()):
        for f in range(t.face_count()):
            im = synth.photo(w, h, seed=40 + k)
            k += 1
            assert t.set_image(im, CubeFace(f), 0, d) if dim == Dimension.Cube else t.set_image(im, 0, d)
    assert t.generate_mipmaps(BOX)
    sources = {}
    for (m, d, f) in _surfaces(t):
        sources[(m, d, f)] = (t.get_image(CubeFace(f), m, d) if dim == Dimension.Cube else t.get_image(m, d)).copy()
    assert t.convert(Format.BC1_RGB, Type.UNorm)
    assert t.converted() and t.get_image(0, 0) is None if dim != Dimension.Cube else t.converted()
    payloads = {}
    for key, im in sources.items():
        m, d, f = key
        got = t.data(CubeFace(f), m, d) if dim == Dimension.Cube else t.data(m, d)
        ref = O.encode(im, int(Format.BC1_RGB), quality=2, threads=2)
        assert np.array_equal(np.asarray(got), ref), key
        assert t.data_size(CubeFace(f), m, d) == ref.size
        payloads[key] = ref.tobytes()
    levels, faces = t.mip_level_count(), t.face_count()
    is3d = dim == Dimension.Dim3D
    # DDS: element -> face -> level -> slice
    res, dds = t.save_bytes(FileType.DDS)
    assert res == SaveResult.Success
    want = b"".join(payloads[(m, v + e, f)] for e in range(t.depth() if t.is_array() else 1)
                    for f in range(faces) for m in range(levels) for v in range(t.depth(m) if is3d else 1))
    assert dds[148:] == want
    assert struct.unpack_from("<I", dds, 128 + 4)[0] == {Dimension.Dim1D: 2, Dimension.Dim2D: 3, Dimension.Cube: 3, Dimension.Dim3D: 4}[dim]
    # PVR: level -> depth -> face (16 bytes of BC1 metadata, + 16 per array / 1-D flag)
    res, pvr = t.save_bytes(FileType.PVR)
    assert res == SaveResult.Success
    meta = struct.unpack_from("<I", pvr, 48)[0]
    assert meta == 16*(1 + int(t.is_array()) + int(dim == Dimension.Dim1D))
    assert pvr[52 + meta:] == b"".join(payloads[(m, d, f)] for m in range(levels) for d in range(t.depth(m)) for f in range(faces))
    # KTX: per level imageSize then depth -> face
    res, ktx = t.save_bytes(FileType.KTX)
    assert res == SaveResult.Success
    assert struct.unpack_from("<I", ktx, 12 + 4*6 + 4)[0] == (0 if dim == Dimension.Dim1D else h)
    off = 64
    for m in range(levels):
        size = struct.unpack_from("<I", ktx, off)[0]
        body = b"".join(payloads[(m, d, f)] for d in range(t.depth(m)) for f in range(faces))
        one_face = dim == Dimension.Cube and not t.is_array()
        assert size == (len(body)//6 if one_face else len(body))
        assert ktx[off + 4:off + 4 + len(body)] == body
        off += 4 + len(body)
    assert off == len(ktx)
    # by file name
    path = str(tmp_path/"out.KTX")
    assert t.save(path) == SaveResult.Success and open(path, "rb").read() == ktx
    assert t.save(str(tmp_path/"out.bin")) == SaveResult.UnknownFormat
    assert t.save(str(tmp_path/"no"/"such"/"dir.dds")) == SaveResult.WriteError
    assert t.save(None) == SaveResult.Invalid


def test_save_results_follow_the_reference_tables():
    """TextureSaveTest.cpp: A8B8G8R8 has no DDS form, R4G4 no KTX form -> Unsupported."""
    img = np.zeros((16, 16, 4), np.float32)
    for fmt, typ, outcomes in ((Format.A8B8G8R8, Type.UNorm, (SaveResult.Unsupported, SaveResult.Success, SaveResult.Success)),
                               (Format.R4G4, Type.UNorm, (SaveResult.Success, SaveResult.Unsupported, SaveResult.Success)),
                               (Format.ASTC_6x6, Type.UNorm, (SaveResult.Unsupported, SaveResult.Success, SaveResult.Success)),
                               (Format.BC7, Type.UNorm, (SaveResult.Success,)*3)):
        t = Texture(Dimension.Dim2D, 16, 16)
        assert t.set_image(img) and t.convert(fmt, typ)
        got = tuple(t.save_bytes(ft)[0] for ft in (FileType.DDS, FileType.KTX, FileType.PVR))
        assert got == outcomes, fmt
        assert t.save_bytes(FileType.Auto)[0] == SaveResult.UnknownFormat
