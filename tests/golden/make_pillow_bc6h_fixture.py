#!/usr/bin/env python3
"""Generate tests/golden/pillow_bc6h.npz: random BC6H blocks (64 per mode, 14 modes, UF16 and
SF16) and the clamped 8-bit RGB Pillow decodes them to.  Data only; pins the bit-field
layouts of oracle/bc6h_decode.c.  Runs only where Pillow is installed."""
import sys

import numpy as np

from make_pillow_fixtures import dds_dx10  # noqa: E402  (same directory)
from PIL import Image
import io

MODES = [(2, 0x00), (2, 0x01), (5, 0x02), (5, 0x06), (5, 0x0A), (5, 0x0E), (5, 0x12), (5, 0x16),
         (5, 0x1A), (5, 0x1E), (5, 0x03), (5, 0x07), (5, 0x0B), (5, 0x0F)]


def decode(dxgi, blocks):
    n = blocks.shape[0]
    im = Image.open(io.BytesIO(dds_dx10(4 * n, 4, dxgi, blocks.tobytes())))
    im.load()
    a = np.asarray(im)[:, :, :3]
    return a.reshape(4, n, 4, 3).transpose(1, 0, 2, 3).reshape(n, 16, 3)


def main(out):
    rng = np.random.default_rng(0xBC6)
    data = {}
    for name, dxgi in (("uf16", 95), ("sf16", 96)):
        blocks = rng.integers(0, 256, size=(64 * 14, 16), dtype=np.uint8)
        for i in range(blocks.shape[0]):
            nb, val = MODES[i // 64]
            b0 = int(blocks[i, 0])
            blocks[i, 0] = (b0 & ~((1 << nb) - 1) & 0xFF) | val
        data[name + "_blocks"] = blocks
        data[name + "_pixels"] = decode(dxgi, blocks)
    np.savez_compressed(out, **data)
    print({k: v.shape for k, v in data.items()})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "tests/golden/pillow_bc6h.npz")
