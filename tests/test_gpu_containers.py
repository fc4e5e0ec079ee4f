"""GPU payload -> DDS -> Pillow (independent reader and decoder) == oracle decode of the same
payload: the end-to-end check that what the HIP kernels write is a valid, correctly laid out
BCn stream."""
import io

import numpy as np
import pytest

import oracle_lib as O
from cuttlefish_amd import Format, Quality, Texture, Type, synth
from cuttlefish_amd import containers as C

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fmt", [Format.BC1_RGB, Format.BC3, Format.BC5, Format.BC7])
def test_gpu_payload_in_dds_decodes_in_pillow(fmt):
    PIL = pytest.importorskip("PIL.Image")
    base = synth.photo(128, 96, seed=11)
    levels, imgs = [], [base]
    for _ in range(4):
        a = imgs[-1].astype(np.uint16)
        imgs.append(((a[0::2, 0::2] + a[1::2, 0::2] + a[0::2, 1::2] + a[1::2, 1::2] + 2) // 4)
                    .astype(np.uint8))
    t = Texture(128, 96, mip_levels=5)
    for i, m in enumerate(imgs):
        assert t.set_image(m, mip=i)
    assert t.convert(fmt, Type.UNorm, Quality.Normal)      # all five levels in one batched launch
    levels = [np.asarray(t.data(i)).copy() for i in range(5)]
    buf = io.BytesIO()
    C.write_dds(buf, fmt, Type.UNorm, 128, 96, levels)
    im = PIL.open(io.BytesIO(buf.getvalue()))
    im.load()
    got = np.asarray(im)
    ref = O.decode(levels[0], int(fmt), 128, 96, 0)
    c = 2 if fmt == Format.BC5 else got.shape[2]
    assert np.array_equal(got[..., :c], ref[..., :c])
    # and the payload itself is the oracle's (same search, byte for byte)
    assert np.array_equal(levels[0], O.encode(base, int(fmt), 0, quality=2, threads=8))


def test_gpu_packed_uncompressed_texture_in_dds_and_pvr(tmp_path):
    """Texture.convert to an uncompressed format (csrc/std_pack.hip) -> DDS -> Pillow."""
    PIL = pytest.importorskip("PIL.Image")
    base = synth.photo(96, 64, seed=12)
    f = base.astype(np.float32)/np.float32(255.0)
    t = Texture(96, 64)
    assert t.set_image(f)
    assert not t.convert(Format.R5G6B5, Type.Float)            # createConverter -> nullptr
    assert t.convert(Format.R8G8B8A8, Type.UNorm)
    assert t.data_size() == 96*64*4
    payload = np.asarray(t.data()).copy()
    assert np.array_equal(payload.reshape(64, 96, 4), base)     # u8 -> float -> UNorm8 round trip
    buf = io.BytesIO()
    C.write_dds(buf, Format.R8G8B8A8, Type.UNorm, 96, 64, [payload])
    got = np.asarray(PIL.open(io.BytesIO(buf.getvalue())).convert("RGBA"))
    assert np.array_equal(got, base)
    buf = io.BytesIO()
    C.write_pvr(buf, Format.R8G8B8A8, Type.UNorm, 96, 64, [payload])
    h = C.read_pvr(buf.getvalue())
    assert buf.getvalue()[h["offset"]:] == payload.tobytes() and h["channel_type"] == 0
