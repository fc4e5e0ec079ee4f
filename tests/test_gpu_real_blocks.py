"""GPU parity on blocks of REAL photographs (tests/golden/real_blocks.npz, see make_real_blocks.py): the kernels'
mode-1 / mode-6 / three-subset / mode-4 paths -- which the synthetic tile hardly visits -- run at volume on the GPU
box, byte for byte against the oracle, at every Texture::Quality."""
import numpy as np
import pytest

import oracle_lib as O
import real_lib as R
from cuttlefish_amd import ColorSpace, Format, Type, make_params

pytestmark = pytest.mark.gpu


def _cmp(ctx, img, fmt, quality, bs, **kw):
    okw = {}
    if "color_space" in kw:
        okw["color_space"] = int(kw["color_space"])
    ref = O.encode(img, int(fmt), quality=quality, threads=8, **okw)
    got = ctx.encode([img], make_params(fmt, Type.UNorm, quality, **kw))[0]
    bad = np.flatnonzero((ref.reshape(-1, bs) != got.reshape(-1, bs)).any(axis=1))
    assert bad.size == 0, "%d blocks differ, first %s" % (bad.size, bad[:8])


@pytest.mark.parametrize("quality", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("alpha", [False, True])
def test_bc7_real_blocks(gpu_ctx, quality, alpha):
    n = 1024 if quality < 4 else 512
    _cmp(gpu_ctx, R.strip(R.blocks4(n, alpha=alpha)), Format.BC7, quality, 16)


@pytest.mark.parametrize("quality", [2, 4])
def test_bc7_real_blocks_srgb(gpu_ctx, quality):
    img = np.concatenate([R.strip(R.blocks4(256)), R.strip(R.blocks4(256, alpha=True))], axis=1)
    _cmp(gpu_ctx, np.ascontiguousarray(img), Format.BC7, quality, 16, color_space=ColorSpace.sRGB)


def test_bc7_real_blocks_odd_pairing(gpu_ctx):
    """an odd number of blocks per strip and mixed opaque / alpha neighbours: every pairing of halves"""
    a, b = R.blocks4(37), R.blocks4(37, alpha=True)
    mixed = np.empty((74, 4, 4, 4), np.uint8)
    mixed[0::2], mixed[1::2] = a, b
    _cmp(gpu_ctx, R.strip(mixed[:73]), Format.BC7, 2, 16)


@pytest.mark.parametrize("fmt,quality", [(Format.BC1_RGB, 2), (Format.BC3, 2), (Format.ETC2_R8G8B8, 2),
                                         (Format.ETC2_R8G8B8, 4), (Format.ETC1, 2)])
def test_4x4_formats_real_blocks(gpu_ctx, fmt, quality):
    bs = O.block_bytes(int(fmt))
    _cmp(gpu_ctx, R.strip(R.blocks4(512)), fmt, quality, bs)


@pytest.mark.parametrize("bw,bh,quality", [(4, 4, 2), (6, 6, 3), (6, 6, 2), (8, 8, 2), (5, 4, 2), (10, 10, 2), (12, 12, 2),
                                          (4, 4, 4), (5, 5, 3), (6, 6, 4), (8, 8, 3), (10, 10, 4), (12, 12, 3), (10, 6, 3), (8, 5, 1)])
def test_astc_real_blocks(gpu_ctx, bw, bh, quality):
    fmt = getattr(Format, "ASTC_%dx%d" % (bw, bh))
    _cmp(gpu_ctx, R.strip(R.blocks(bw, bh, 256)), fmt, quality, 16)
