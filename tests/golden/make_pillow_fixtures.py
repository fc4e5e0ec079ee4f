#!/usr/bin/env python3
"""Generate tests/golden/pillow_decode.npz: random + structured BCn blocks and the
pixels Pillow's independent BCn decoder produces for them.

Runs only where Pillow is installed (the build container).  The committed .npz
is data only (block bytes in, decoded pixels out); it pins oracle/bcn_decode.c
("What pins our results instead").  No reference code is
involved: the reference (Cuttlefish) has no decoder and its codecs are absent.
"""
import io
import struct
import sys

import numpy as np
from PIL import Image

DXGI = {"bc1": 71, "bc2": 74, "bc3": 77, "bc4u": 80, "bc5u": 83, "bc5s": 84, "bc7": 98}
BYTES = {"bc1": 8, "bc2": 16, "bc3": 16, "bc4u": 8, "bc5u": 16, "bc5s": 16, "bc7": 16}


def dds_dx10(w, h, dxgi, payload):
    hdr = struct.pack("<4sI I II I I I 11I", b"DDS ", 124, 0x1 | 0x2 | 0x4 | 0x1000 | 0x80000,
                      h, w, len(payload), 0, 1, *([0] * 11))
    pf = struct.pack("<II4sIIIII", 32, 0x4, b"DX10", 0, 0, 0, 0, 0)
    caps = struct.pack("<IIIII", 0x1000, 0, 0, 0, 0)
    dx10 = struct.pack("<IIIII", dxgi, 3, 0, 1, 0)
    return hdr + pf + caps + dx10 + payload


def decode(fmt, blocks):
    """blocks: (n, bytes) uint8 -> (n, 16, channels) via one n*4 x 4 image."""
    n = blocks.shape[0]
    im = Image.open(io.BytesIO(dds_dx10(4 * n, 4, DXGI[fmt], blocks.tobytes())))
    im.load()
    a = np.asarray(im)
    if a.ndim == 2:
        a = a[:, :, None]
    # (4, 4n, c) -> (n, 16, c)
    a = a.reshape(4, n, 4, a.shape[2]).transpose(1, 0, 2, 3).reshape(n, 16, a.shape[2])
    return a, im.mode


def main(out):
    rng = np.random.default_rng(0xC0FFEE)
    data = {}
    for fmt in DXGI:
        n = 512 if fmt == "bc7" else 256
        blocks = rng.integers(0, 256, size=(n, BYTES[fmt]), dtype=np.uint8)
        if fmt == "bc7":
            # 64 blocks per mode: mode m = m zero bits then a one
            for i in range(n):
                m = i // 64
                b0 = int(blocks[i, 0])
                blocks[i, 0] = ((b0 & ~((1 << (m + 1)) - 1)) & 0xFF) | (1 << m)
        if fmt in ("bc1", "bc2", "bc3"):
            # include equal-endpoint and ordered/unordered endpoint cases
            off = 0 if fmt == "bc1" else 8
            blocks[0:8, off + 2:off + 4] = blocks[0:8, off:off + 2]
        if fmt in ("bc4u", "bc5u", "bc5s"):
            blocks[0:8, 1] = blocks[0:8, 0]
            blocks[8, 0:2] = (0x80, 0x7F)     # snorm -128 endpoint
            blocks[9, 0:2] = (0x7F, 0x80)
            blocks[10, 0:2] = (0x80, 0x80)
        px, mode = decode(fmt, blocks)
        data[fmt + "_blocks"] = blocks
        data[fmt + "_pixels"] = px
        print(fmt, "PIL mode", mode, "pixels", px.shape, px.dtype)
    np.savez_compressed(out, **data)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "tests/golden/pillow_decode.npz")
