"""GPU mip-chain generation (cfhip_generate_mips_device) vs the oracle restatement of
Texture::generateMipmaps / Image::resize.

Tolerance (floating-point path): the linear-colour-space resize is pure double add / divide
with float stores -> bit-exact.  The sRGB round trip goes through pow(): ocml (GPU) and libm
(CPU) may differ in the last bit of the double, which can move the float store by 1 ulp; the
test allows 2 float ulps per level-to-level step and reports how many texels differ at all.
"""
import numpy as np
import pytest

import oracle_lib as O
from cuttlefish_amd import CfhipError, ColorSpace, Context, Format, PixelType, Type, make_params, \
    payload_size, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _gpu_chain(ctx, base, levels, color_space, filt):
    h, w = base.shape[:2]
    src = torch.from_numpy(np.ascontiguousarray(base)).cuda()
    if base.dtype == np.uint8:
        pt, pitch = PixelType.RGBA8, w*4
    elif base.dtype == np.float16:
        pt, pitch = PixelType.RGBA16F, w*8
    else:
        pt, pitch = PixelType.RGBA32F, w*16
    dsts = [torch.empty((max(1, h >> k), max(1, w >> k), 4), dtype=torch.float32, device="cuda")
            for k in range(1, levels)]
    ctx.generate_mips_device(src.data_ptr(), pt, w, h, pitch, [d.data_ptr() for d in dsts],
                             color_space=color_space, filter=filt)
    return [d.cpu().numpy() for d in dsts], dsts, src


def _ulps(a, b):
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    return np.abs(ai - bi)


@pytest.mark.parametrize("filt", [0, 1, 2, 3, 4, 0 | O.FILTER_FALLBACK, 1 | O.FILTER_FALLBACK])
def test_linear_space_chain_is_bit_exact(filt):
    """All five filters through the FreeImage_Rescale passes (what a stock build computes,
    Image.cpp:1348-1380) and Box / Linear through the in-tree fallback loops (:1393-1505)."""
    rng = np.random.default_rng(1)
    base = rng.random((192, 256, 4)).astype(np.float32)
    with Context(0) as ctx:
        got, _, _ = _gpu_chain(ctx, base, 9, ColorSpace.Linear, filt)
    ref = O.mip_chain(base, 9, filter=filt, color_space=0)[1:]
    for g, r in zip(got, ref):
        assert g.shape == r.shape
        assert np.array_equal(g, r)


@pytest.mark.parametrize("dtype", [np.uint8, np.float16])
def test_base_level_pixel_types(dtype):
    img = synth.photo(96, 64, seed=2)
    base = img if dtype == np.uint8 else (img.astype(np.float32)/255.0).astype(np.float16)
    with Context(0) as ctx:
        got, _, _ = _gpu_chain(ctx, base, 7, ColorSpace.Linear, 0)
    ref = O.mip_chain(base, 7)[1:]
    for g, r in zip(got, ref):
        assert np.array_equal(g, r)


@pytest.mark.parametrize("filt", [0, 3, 0 | O.FILTER_FALLBACK])
def test_srgb_chain_within_pow_rounding(filt):
    img = synth.photo(256, 256, seed=4)
    with Context(0) as ctx:
        got, _, _ = _gpu_chain(ctx, img, 9, ColorSpace.sRGB, filt)
        # level k from the GPU's own level k-1 on the CPU: isolates one resize step
        prev = (img.astype(np.float64)/255.0).astype(np.float32)
        worst, differing, total = 0, 0, 0
        for k, g in enumerate(got, start=1):
            r = O.resize_rgbaf(prev, g.shape[1], g.shape[0], filt, 1)
            u = _ulps(g, r)
            worst = max(worst, int(u.max()))
            differing += int((u > 0).sum())
            total += u.size
            assert np.array_equal(g[..., 3], r[..., 3])          # alpha never sees pow()
            prev_ok = True
            prev = g
    assert worst <= 2, worst
    assert differing <= total*0.02, (differing, total)


def test_unsupported_filter_and_bad_arguments():
    x = torch.zeros((8, 8, 4), dtype=torch.float32, device="cuda")
    d = torch.zeros((4, 4, 4), dtype=torch.float32, device="cuda")
    with Context(0) as ctx:
        with pytest.raises(CfhipError) as e:
            ctx.generate_mips_device(x.data_ptr(), PixelType.RGBA32F, 8, 8, 128, [d.data_ptr()], filter=9)
        assert e.value.code == -1                                # not a ResizeFilter
        with pytest.raises(CfhipError) as e:
            ctx.generate_mips_device(x.data_ptr(), PixelType.RGBA32F, 8, 8, 128, [d.data_ptr()],
                                     filter=3 | O.FILTER_FALLBACK)
        assert e.value.code == -1                                # the fallback knows Box and Linear only
        with pytest.raises(CfhipError) as e:
            ctx.generate_mips_device(x.data_ptr(), PixelType.RGBA32F, 8, 8, 64, [d.data_ptr()])
        assert e.value.code == -1                                # pitch smaller than a row
        with pytest.raises(CfhipError):
            ctx.generate_mips_device(x.data_ptr(), PixelType.RGBA32F, 8, 8, 128, [d.data_ptr()]*6)   # > 4 levels


def test_mips_feed_the_encoder_without_leaving_the_gpu():
    img = synth.photo(512, 512, seed=6)
    fmt, typ = Format.BC7, Type.UNorm
    with Context(0) as ctx:
        got, dsts, src = _gpu_chain(ctx, img, 8, ColorSpace.Linear, 0)
        p = make_params(fmt, typ, 2)
        surfaces, outs = [], []
        levels = [(src, PixelType.RGBA8, 512, 512, 512*4)] + \
            [(d, PixelType.RGBA32F, d.shape[1], d.shape[0], d.shape[1]*16) for d in dsts]
        for t, pt, w, h, pitch in levels:
            o = torch.empty(payload_size(fmt, typ, w, h), dtype=torch.uint8, device="cuda")
            outs.append(o)
            surfaces.append({"pixels": t.data_ptr(), "pixel_type": pt, "width": w, "height": h,
                             "row_pitch_bytes": pitch, "out": o.data_ptr(), "out_capacity": o.numel()})
        ctx.encode_device(surfaces, p)                           # one batched launch, all 8 levels
        torch.cuda.synchronize()
    ref_chain = O.mip_chain(img, 8)
    for k, o in enumerate(outs):
        src_k = img if k == 0 else ref_chain[k]                  # linear space: GPU mips == oracle mips
        assert np.array_equal(o.cpu().numpy(), O.encode(src_k, int(fmt), 0, quality=2, threads=8))


def test_texture_generate_mipmaps_mirror():
    from cuttlefish_amd import Quality, ResizeFilter, Texture
    img = synth.photo(64, 32, seed=8)
    t = Texture(64, 32)
    assert t.set_image(img)
    assert t.generate_mipmaps(ResizeFilter.Box)
    assert t.mip_level_count() == 7                      # 64x32 .. 1x1
    assert t.convert(Format.BC7, Type.UNorm, Quality.Low)
    ref = O.mip_chain(img, 7)
    for k in range(7):
        src_k = img if k == 0 else ref[k]
        assert np.array_equal(np.asarray(t.data(k)), O.encode(src_k, int(Format.BC7), 0, quality=1))


def test_default_catmull_rom_chain_through_the_texture_mirror():
    from cuttlefish_amd import Quality, Texture
    img = synth.photo(96, 64, seed=9)
    t = Texture(96, 64)
    assert t.set_image(img)
    assert t.generate_mipmaps()                          # Image::ResizeFilter::CatmullRom, all levels
    assert t.mip_level_count() == 7
    assert t.convert(Format.BC1_RGB, Type.UNorm, Quality.Normal)
    ref = O.mip_chain(img, 7, filter=3)
    for k in range(7):
        src_k = img if k == 0 else ref[k]
        assert np.array_equal(np.asarray(t.data(k)), O.encode(src_k, int(Format.BC1_RGB), 0, quality=2))


def test_generate_mipmaps_level_dimensions_as_in_the_reference_test():
    """TextureTest.GenerateMipmaps (lib/test/TextureTest.cpp:539-568): a 15x10 texture gets 4
    levels of 15x10, 7x5, 3x2 and 1x1 (2-D form of the reference's cube-array case); an image of
    the wrong size is refused."""
    from cuttlefish_amd import Texture
    t = Texture(15, 10)
    assert not t.set_image(np.zeros((15, 10, 4), np.float32))       # 10 wide, 15 high: wrong size
    assert not t.images_complete()
    assert t.set_image(np.zeros((10, 15, 4), np.float32))
    assert t.images_complete() and t.generate_mipmaps() and t.images_complete()
    assert t.mip_level_count() == 4
    assert [t.get_image(k).shape[:2] for k in range(4)] == [(10, 15), (5, 7), (2, 3), (1, 1)]


def _gpu_chain3d(ctx, vol, levels, color_space, filt):
    d, h, w = vol.shape[:3]
    src = torch.from_numpy(np.ascontiguousarray(vol)).cuda()
    if vol.dtype == np.uint8:
        pt, texel = PixelType.RGBA8, 4
    else:
        pt, texel = PixelType.RGBA32F, 16
    dsts = [torch.empty((max(1, d >> k), max(1, h >> k), max(1, w >> k), 4), dtype=torch.float32, device="cuda")
            for k in range(1, levels)]
    ctx.generate_mips3d_device(src.data_ptr(), pt, w, h, d, w*texel, w*h*texel, [t.data_ptr() for t in dsts],
                               color_space=color_space, filter=filt)
    return [t.cpu().numpy() for t in dsts]


@pytest.mark.parametrize("filt", [0, 1, 3])
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_3d_chain_linear_space_is_bit_exact(filt, dtype):
    """Texture::generateMipmaps for Dim3D (Texture.cpp:1345-1440): x / y resize of every slice, then
    generateMips3d along the depth; non-power-of-two sizes, down to 1x1x1."""
    rng = np.random.default_rng(11)
    vol = (rng.random((12, 20, 28, 4))*255).astype(np.uint8)
    if dtype == np.float32:
        vol = (vol.astype(np.float32)/255.0)*np.float32(1.5)
    with Context(0) as ctx:
        got = _gpu_chain3d(ctx, vol, 5, ColorSpace.Linear, filt)
    want = O.mip_chain3d(vol, 5, filter=filt, color_space=0)[1:]
    assert [g.shape for g in got] == [w.shape for w in want]
    for k, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g, w), "level %d differs" % (k + 1)


def test_3d_chain_srgb_within_pow_rounding():
    rng = np.random.default_rng(12)
    vol = (rng.random((8, 16, 16, 4))*255).astype(np.uint8)
    with Context(0) as ctx:
        got = _gpu_chain3d(ctx, vol, 4, ColorSpace.sRGB, 0)
    want = O.mip_chain3d(vol, 4, filter=0, color_space=1)[1:]
    for k, (g, w) in enumerate(zip(got, want)):
        # two pow round trips per level (x / y resize, depth pass): 2 ulps each, accumulated over levels
        assert _ulps(g, w).max() <= 4*(k + 1), "level %d" % (k + 1)
        assert np.array_equal(g[..., 3], w[..., 3])


def test_3d_bad_arguments():
    with Context(0) as ctx:
        t = torch.zeros((4, 4, 4, 4), dtype=torch.uint8, device="cuda")
        o = torch.zeros((2, 2, 2, 4), dtype=torch.float32, device="cuda")
        with pytest.raises(CfhipError):
            ctx.generate_mips3d_device(t.data_ptr(), PixelType.RGBA8, 4, 4, 4, 16, 64, [o.data_ptr()]*3)   # 4 levels of a 4^3
        with pytest.raises(CfhipError):
            ctx.generate_mips3d_device(t.data_ptr(), PixelType.RGBA8, 4, 4, 4, 16, 32, [o.data_ptr()])     # slice pitch < slice


@pytest.mark.parametrize("filt,cs,shape,dtype", [
    (0, ColorSpace.Linear, (128, 128), np.uint8), (3, ColorSpace.Linear, (37, 20), np.float32),
    (1, ColorSpace.sRGB, (64, 96), np.uint8), (0 | O.FILTER_FALLBACK, ColorSpace.Linear, (48, 48), np.float16),
    (4, ColorSpace.sRGB, (33, 1), np.float32)])
def test_array_layers_in_one_call_equal_one_call_per_layer(filt, cs, shape, dtype):
    """cfhip_generate_mips_array_device: the layers of an array texture share one launch per pass and
    level (Texture::generateMipmaps resizes every [depth][face] image on its own); every level of every
    layer is bit-identical to the per-layer call, on the default stream and on a stream of the caller's."""
    h, w = shape
    rng = np.random.default_rng(7)
    layers = 5
    levels = max(h, w).bit_length()
    bases = []
    for l in range(layers):
        f = rng.random((h, w, 4)).astype(np.float32)*(l + 1)/layers
        bases.append((f*255).astype(np.uint8) if dtype == np.uint8 else f.astype(dtype))
    with Context(0) as ctx:
        want = [_gpu_chain(ctx, b, levels, cs, filt)[0] for b in bases]
        for use_stream in (False, True):
            srcs = [torch.from_numpy(np.ascontiguousarray(b)).cuda() for b in bases]
            pt = {np.uint8: PixelType.RGBA8, np.float16: PixelType.RGBA16F, np.float32: PixelType.RGBA32F}[dtype]
            pitch = w*{np.uint8: 4, np.float16: 8, np.float32: 16}[dtype]
            dsts = [[torch.zeros((max(1, h >> k), max(1, w >> k), 4), dtype=torch.float32, device="cuda")
                     for k in range(1, levels)] for _ in range(layers)]
            torch.cuda.synchronize()
            st = torch.cuda.Stream() if use_stream else None
            ctx.generate_mips_array_device([s.data_ptr() for s in srcs], pt, w, h, pitch,
                                           [[d.data_ptr() for d in dl] for dl in dsts], color_space=cs, filter=filt,
                                           stream=st.cuda_stream if st else 0)
            torch.cuda.synchronize()
            for l in range(layers):
                for k in range(levels - 1):
                    assert np.array_equal(dsts[l][k].cpu().numpy(), want[l][k]), (use_stream, l, k)


def test_array_call_rejects_bad_arguments():
    with Context(0) as ctx:
        a = torch.zeros((8, 8, 4), dtype=torch.uint8, device="cuda")
        d = torch.zeros((4, 4, 4), dtype=torch.float32, device="cuda")
        with pytest.raises(ValueError):
            ctx.generate_mips_array_device([a.data_ptr(), a.data_ptr()], PixelType.RGBA8, 8, 8, 32, [[d.data_ptr()]])
        with pytest.raises(CfhipError):
            ctx.generate_mips_array_device([a.data_ptr()], PixelType.RGBA8, 8, 8, 32, [[0]])          # NULL level
        with pytest.raises(CfhipError):
            ctx.generate_mips_array_device([a.data_ptr()], PixelType.RGBA8, 8, 8, 16, [[d.data_ptr()]])   # pitch < row
