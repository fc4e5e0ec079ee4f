"""GPU parity for BC6H (UF16 / SF16): HIP kernel through the C-ABI vs the CPU oracle."""
import numpy as np
import pytest

import oracle_lib as O
from cuttlefish_amd import Format, Texture, Type, make_params, synth

pytestmark = pytest.mark.gpu
BC6H = int(Format.BC6H)


def _gpu(ctx, img, typ, quality=2):
    return ctx.encode([img], make_params(Format.BC6H, typ, quality))[0]


@pytest.mark.parametrize("signed,typ", [(False, Type.UFloat), (True, Type.Float)])
@pytest.mark.parametrize("quality", [0, 1, 2, 3, 4])
def test_bit_exact_vs_oracle_half_source(gpu_ctx, signed, typ, quality):
    img = synth.hdr_probe(64, 48, seed=30 + quality, signed=signed)
    ref = O.encode(img, BC6H, typ=int(typ), quality=quality, threads=8)
    got = _gpu(gpu_ctx, img, typ, quality)
    bad = np.flatnonzero((ref.reshape(-1, 16) != got.reshape(-1, 16)).any(axis=1))
    assert bad.size == 0, "blocks differ: %s" % bad[:10]


def test_float32_source_rne_packing(gpu_ctx):
    """fp32 -> fp16 RNE on the GPU (v_cvt_f16_f32) == packHardwareHalfFloat (HalfFloat.h:96-136)."""
    rng = np.random.default_rng(3)
    f = (rng.standard_normal((40, 36, 4)) * 10.0 ** rng.integers(-4, 5, (40, 36, 4))).astype(np.float32)
    for typ in (Type.UFloat, Type.Float):
        ref = O.encode(f, BC6H, typ=int(typ), quality=2, threads=8)
        assert np.array_equal(ref, _gpu(gpu_ctx, f, typ))


def test_rgba8_source_and_ragged_sizes(gpu_ctx):
    for w, h in [(1, 1), (7, 5), (33, 18)]:
        img = synth.photo(w, h, seed=w * 7 + h)
        ref = O.encode(img, BC6H, typ=int(Type.UFloat), quality=1, threads=4)
        assert np.array_equal(ref, _gpu(gpu_ctx, img, Type.UFloat, 1))
        hdr = synth.hdr_probe(w, h, seed=w + h)
        ref = O.encode(hdr, BC6H, typ=int(Type.UFloat), quality=2, threads=4)
        assert np.array_equal(ref, _gpu(gpu_ctx, hdr, Type.UFloat, 2))


def test_texture_convert_size_contract(gpu_ctx):
    t = Texture(16, 16)
    img = np.zeros((16, 16, 4), np.float32)
    img[..., 3] = 1.0
    assert t.set_image(img)
    assert not t.convert(Format.BC6H, Type.UNorm)        # createConverter -> nullptr
    assert t.convert(Format.BC6H, Type.UFloat)
    assert t.data_size() == 4 * 4 * 16


def test_config4_full_size_properties_2048(gpu_ctx):
    """BASELINE config 4: 2048x2048 RGBA16F HDR probe, BC6H UFLOAT."""
    img = synth.hdr_probe(2048, 2048, seed=4)
    a = _gpu(gpu_ctx, img, Type.UFloat, 2)
    assert a.nbytes == 512 * 512 * 16
    assert np.array_equal(a, _gpu(gpu_ctx, img, Type.UFloat, 2))
    dec = O.decode_bc6h(a, 2048, 2048, int(Type.UFloat))
    assert synth.psnr_log(img, dec) > 50.0
    strip = img[1024:1040]
    ref = O.encode(strip, BC6H, typ=int(Type.UFloat), quality=2, threads=8)
    assert np.array_equal(ref, a.reshape(512, 512 * 16)[256:260].reshape(-1))
