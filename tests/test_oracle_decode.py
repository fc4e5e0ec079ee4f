"""Pin oracle/bcn_decode.c bit-for-bit to Pillow's independent BCn decoder through the
committed fixture tests/golden/pillow_decode.npz (made by make_pillow_fixtures.py)."""
import ctypes
import os

import numpy as np
import pytest

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(__file__), "golden", "pillow_decode.npz")


def _run(fn, blocks, outn, dtype=np.uint8):
    L = O.lib()
    out = np.zeros((blocks.shape[0], outn), dtype)
    for i in range(blocks.shape[0]):
        b = np.ascontiguousarray(blocks[i])
        getattr(L, fn)(ctypes.c_void_p(b.ctypes.data), ctypes.c_void_p(out[i].ctypes.data))
    return out


@pytest.mark.parametrize("fmt,fn", [("bc1", "cfo_decode_bc1"), ("bc2", "cfo_decode_bc2"),
                                    ("bc3", "cfo_decode_bc3"), ("bc7", "cfo_decode_bc7")])
def test_rgba_decoders_match_pillow(fmt, fn):
    d = np.load(GOLD)
    got = _run(fn, d[fmt + "_blocks"], 64).reshape(-1, 16, 4)
    assert np.array_equal(got, d[fmt + "_pixels"])


def test_bc7_fixture_covers_all_modes():
    d = np.load(GOLD)
    b0 = d["bc7_blocks"][:, 0]
    modes = set()
    for v in b0:
        m = 0
        while not (int(v) >> m) & 1:
            m += 1
        modes.add(m)
    assert modes == set(range(8))


def test_bc4_bc5_unorm_match_pillow():
    d = np.load(GOLD)
    assert np.array_equal(_run("cfo_decode_bc4u", d["bc4u_blocks"], 16), d["bc4u_pixels"][:, :, 0])
    b = d["bc5u_blocks"]
    assert np.array_equal(_run("cfo_decode_bc4u", b[:, :8], 16), d["bc5u_pixels"][:, :, 0])
    assert np.array_equal(_run("cfo_decode_bc4u", b[:, 8:], 16), d["bc5u_pixels"][:, :, 1])


def test_bc5_snorm_matches_pillow_except_minus128_convention():
    """Pillow keeps -128 (and uses it as the explicit minimum of the 6-value mode); D3D
    clamps it to -127.  Outside those cases the decoders must agree exactly."""
    d = np.load(GOLD)
    B, P = d["bc5s_blocks"], d["bc5s_pixels"]
    checked = 0
    for half in (0, 1):
        blocks = np.ascontiguousarray(B[:, 8 * half:8 * half + 8])
        got = _run("cfo_decode_bc4s", blocks, 16, np.int8).astype(int) + 128
        for i in range(blocks.shape[0]):
            a0, a1 = int(np.int8(blocks[i, 0])), int(np.int8(blocks[i, 1]))
            sel = int.from_bytes(bytes(blocks[i, 2:8]), "little")
            sels = [(sel >> (3 * k)) & 7 for k in range(16)]
            if a0 == -128 or a1 == -128 or (a0 <= a1 and 6 in sels):
                continue
            checked += 1
            assert np.array_equal(got[i], P[i, :, half].astype(int))
    assert checked > 200


def test_decode_image_layout_and_edge_crop():
    """cfo_decode walks blocks row-major and crops partial edge blocks."""
    d = np.load(GOLD)
    blocks = d["bc7_blocks"][:6]          # 3 x 2 blocks -> 10 x 7 image (cropped)
    img = O.decode(blocks.reshape(-1), 36, 10, 7)
    px = d["bc7_pixels"][:6].reshape(6, 4, 4, 4)
    for by in range(2):
        for bx in range(3):
            h = min(4, 7 - by * 4)
            w = min(4, 10 - bx * 4)
            assert np.array_equal(img[by * 4:by * 4 + h, bx * 4:bx * 4 + w],
                                  px[by * 3 + bx][:h, :w])
