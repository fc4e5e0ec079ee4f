"""Generates tests/golden/mesa_astc.npz: for each of the 14 ASTC footprints, 192 structured random
blocks (legal block mode for the footprint, 1-4 partitions, LDR and a few HDR endpoint modes,
mixed per-partition modes, dual plane, every trit / quint / bit range, a few void-extent blocks
and a share of ILLEGAL encodings) with the RGBA8 pixels Mesa 23.2.1's software ASTC decoder
(LDR profile) produced for them -- an independent decoder, see make_mesa_fixtures.py.

    python tests/golden/make_mesa_astc_fixture.py
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import mesa_lib as M  # noqa: E402
import oracle_lib as O  # noqa: E402

FP = [(4, 4), (5, 4), (5, 5), (6, 5), (6, 6), (8, 5), (8, 6), (8, 8), (10, 5), (10, 6), (10, 8),
      (10, 10), (12, 10), (12, 12)]
KEEP_VALID, KEEP_INVALID = 160, 32
MAGENTA = np.array([255, 0, 255, 255], np.uint8)


def legal_modes(bw, bh):
    L = O.lib()
    L.astc_parse_block_mode.argtypes = [ctypes.c_int] + [ctypes.POINTER(ctypes.c_int)]*4
    out = []
    for m in range(2048):
        v = [ctypes.c_int() for _ in range(4)]
        if L.astc_parse_block_mode(m, *v) == 0 and v[0].value <= bw and v[1].value <= bh:
            out.append(m)
    return np.array(out, np.uint64)


def candidates(bw, bh, n, rng):
    vm = legal_modes(bw, bh)
    blk = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    v = blk.view(np.uint64).reshape(n, 2).copy()
    lo = v[:, 0]
    mode = vm[rng.integers(0, len(vm), n)]
    parts = rng.choice(np.array([0, 0, 0, 1, 1, 2, 3], np.uint64), n)
    lo = (lo & ~np.uint64(0x1FFF)) | mode | (parts << np.uint64(11))
    ldr = np.array([0, 1, 4, 5, 6, 8, 9, 10, 12, 13], np.uint64)
    cem = ldr[rng.integers(0, len(ldr), n)]
    single = parts == 0
    lo = np.where(single, (lo & ~np.uint64(0xF << 13)) | (cem << np.uint64(13)), lo)
    same = (~single) & (rng.random(n) < 0.5)          # the other half: per-partition modes, HDR included
    lo = np.where(same, (lo & ~np.uint64(0x3F << 23)) | (cem << np.uint64(25)), lo)
    v[:, 0] = lo
    out = v.view(np.uint8).reshape(n, 16)
    for i in range(0, n, 64):                          # void-extent blocks, no extent coordinates
        out[i, :8] = [0xFC, 0xFD, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF]
    return out


def main():
    assert M.available()
    rng = np.random.default_rng(0xA57C)
    out = {"mesa_version": np.array(M.version())}
    for fi, (bw, bh) in enumerate(FP):
        n = 2048
        blk = candidates(bw, bh, n, rng)
        px = M.decode(43 + fi, blk.reshape(-1), bw*32, bh*(n//32))
        px = px.reshape(n//32, bh, 32, bw, 4).transpose(0, 2, 1, 3, 4).reshape(n, bh*bw, 4)
        bad = (px == MAGENTA).all(-1).any(-1)          # a texel of Mesa's error colour
        keep = np.concatenate([np.nonzero(~bad)[0][:KEEP_VALID], np.nonzero(bad)[0][:KEEP_INVALID]])
        assert (~bad).sum() >= KEEP_VALID
        out["blocks_%dx%d" % (bw, bh)] = blk[keep]
        out["rgba_%dx%d" % (bw, bh)] = px[keep]
    np.savez_compressed(os.path.join(HERE, "mesa_astc.npz"), **out)
    print("wrote mesa_astc.npz")


if __name__ == "__main__":
    main()
