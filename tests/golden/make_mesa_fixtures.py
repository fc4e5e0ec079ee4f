"""Generates tests/golden/mesa_blocks.npz: random (= every mode, every field value) blocks of the
formats whose decode is bit-specified, with the pixels an INDEPENDENT decoder produced for them:
Mesa 23.2.1's software texture decompression behind an off-screen llvmpipe context
(tools/mesa_ref/mesa_decode.c, tests/mesa_lib.py).  Runs only where the Mesa driver file exists
(this image); the committed .npz is what tests/test_oracle_mesa.py checks the oracle decoders
against anywhere.

    python tests/golden/make_mesa_fixtures.py

ETC2 / EAC: every 64- / 128-bit pattern is a valid block (individual, differential, T, H and
planar are selected by overflow), so uniform random blocks cover all modes.  BC6H / BC7: random
blocks hit all 14 / 8 mode layouts (reserved BC6H modes decode to zero).  S3TC / RGTC are NOT in
the fixture: their 1/3- and 1/7-point rounding is implementation-defined (Mesa truncates where
Pillow, which pins those decoders, rounds).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import mesa_lib as M  # noqa: E402

N = 512             # blocks per format
W, H = 128, 4 * N // 32


def main():
    assert M.available(), "Mesa swrast driver not present"
    rng = np.random.default_rng(0xE7C2)
    out = {"mesa_version": np.array(M.version())}
    for name, fmt, bs in (("etc2_rgb", 38, 8), ("etc2_rgba1", 39, 8), ("etc2_rgba8", 40, 16),
                          ("bc7", 36, 16)):
        blk = rng.integers(0, 256, N * bs, dtype=np.uint8)
        out[name + "_blocks"] = blk
        out[name + "_rgba"] = M.decode(fmt, blk, W, H)
    for name, fmt, bs, nch in (("eac_r11", 41, 8, 1), ("eac_rg11", 42, 16, 2)):
        for typ, tn in ((0, "u"), (1, "s")):
            blk = rng.integers(0, 256, N * bs, dtype=np.uint8)
            out["%s_%s_blocks" % (name, tn)] = blk
            # Mesa decodes EAC to 16-bit (un)signed normalised: read back without conversion
            out["%s_%s_px16" % (name, tn)] = M.decode(
                fmt, blk, W, H, typ, rb_format=M.GL_RG if nch == 2 else M.GL_RED,
                rb_type=M.GL_SHORT if typ else M.GL_UNSIGNED_SHORT)
    for typ, tn in ((4, "uf16"), (5, "sf16")):
        blk = rng.integers(0, 256, N * 16, dtype=np.uint8)
        out["bc6h_%s_blocks" % tn] = blk
        out["bc6h_%s_half" % tn] = M.decode(35, blk, W, H, typ, rb_format=M.GL_RGB,
                                            rb_type=M.GL_HALF_FLOAT).view(np.uint16)
    np.savez_compressed(os.path.join(HERE, "mesa_blocks.npz"), **out)
    print("wrote mesa_blocks.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
