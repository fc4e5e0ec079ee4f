#!/usr/bin/env python3
"""Blocks sampled from REAL photographs that sit in the build container -> tests/golden/real_blocks.npz.

The images themselves never travel (same pattern as the Pillow / Mesa fixtures): this script runs HERE, reads the
sample photographs that Python packages of this image ship (sklearn, matplotlib, the conda skimage / imageio data
directories), and commits only a few thousand sampled texel blocks -- data, not pictures.  The quality ladders of
every codec (tools/quality_tables.py --real, tools/bc7_lab.py --content real, tests/test_oracle_bounds.py) are
measured on these blocks; the synthetic generator of cuttlefish_amd/synth.py stays what the throughput bench runs on.

    python tests/golden/make_real_blocks.py          # writes real_blocks.npz next to this file

Content of the archive (all uint8):
    names      the image each block came from (index into `images`)
    rgb4       (N4, 4, 4, 3)      4x4 blocks, opaque photographs
    rgb4_img   (N4,)              source image of each
    rgba4      (NA, 4, 4, 4)      4x4 blocks with an alpha channel that varies: half are blocks of the two RGBA
                                  pictures skimage ships (logo.png, horse.png) that hold a partly transparent texel,
                                  half take the luma of ANOTHER photograph as alpha (smooth + textured alpha)
    rgb12      (N12, 12, 12, 3)   12x12 patches; every ASTC footprint crops its block from the top-left corner
    rgb12_img  (N12,)
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SK = "/opt/conda/lib/python3.9/site-packages/skimage/data/"
PHOTOS = [
    ("china", "/usr/local/lib/python3.10/dist-packages/sklearn/datasets/images/china.jpg"),
    ("flower", "/usr/local/lib/python3.10/dist-packages/sklearn/datasets/images/flower.jpg"),
    ("grace_hopper", "/usr/local/lib/python3.10/dist-packages/matplotlib/mpl-data/sample_data/grace_hopper.jpg"),
    ("astronaut", SK + "astronaut.png"),
    ("coffee", SK + "coffee.png"),
    ("chelsea", SK + "chelsea.png"),
    ("rocket", SK + "rocket.jpg"),
    ("motorcycle", SK + "motorcycle_left.png"),
]
ALPHA_PICS = [("logo", SK + "logo.png"), ("horse", SK + "horse.png")]
N4_PER_IMAGE = 512
NA = 1024
N12_PER_IMAGE = 96


def load(path, mode):
    return np.asarray(Image.open(path).convert(mode))


def main():
    rng = np.random.default_rng(0xC0FFEE)
    photos = [load(p, "RGB") for _, p in PHOTOS]
    rgb4, rgb4_img, rgb12, rgb12_img = [], [], [], []
    for k, img in enumerate(photos):
        h, w = img.shape[:2]
        for _ in range(N4_PER_IMAGE):
            y, x = int(rng.integers(0, h // 4)) * 4, int(rng.integers(0, w // 4)) * 4
            rgb4.append(img[y:y + 4, x:x + 4])
            rgb4_img.append(k)
        for _ in range(N12_PER_IMAGE):
            y, x = int(rng.integers(0, h // 12)) * 12, int(rng.integers(0, w // 12)) * 12
            rgb12.append(img[y:y + 12, x:x + 12])
            rgb12_img.append(k)
    rgba4 = []
    # (a) real alpha: blocks of the RGBA pictures that hold at least one texel with 0 < alpha < 255 or a mix
    pool = []
    for _, p in ALPHA_PICS:
        img = load(p, "RGBA")
        h, w = img.shape[:2]
        for y in range(0, h - 3, 4):
            for x in range(0, w - 3, 4):
                a = img[y:y + 4, x:x + 4, 3]
                if a.min() != a.max():
                    pool.append(img[y:y + 4, x:x + 4])
    pool = np.stack(pool)
    take = rng.permutation(len(pool))[:NA // 2]
    rgba4.extend(pool[take])
    # (b) a photograph's colour with another photograph's luma as alpha
    while len(rgba4) < NA:
        a, b = rng.choice(len(photos), 2, replace=False)
        ia, ib = photos[a], photos[b]
        h, w = min(ia.shape[0], ib.shape[0]), min(ia.shape[1], ib.shape[1])
        y, x = int(rng.integers(0, h // 4)) * 4, int(rng.integers(0, w // 4)) * 4
        lum = (ib[y:y + 4, x:x + 4].astype(np.uint32) @ np.array([54, 183, 19], np.uint32) >> 8).astype(np.uint8)
        rgba4.append(np.dstack([ia[y:y + 4, x:x + 4], lum]))
    out = os.path.join(HERE, "real_blocks.npz")
    np.savez_compressed(
        out, images=np.array([n for n, _ in PHOTOS]),
        rgb4=np.stack(rgb4).astype(np.uint8), rgb4_img=np.array(rgb4_img, np.uint8),
        rgba4=np.stack(rgba4).astype(np.uint8),
        rgb12=np.stack(rgb12).astype(np.uint8), rgb12_img=np.array(rgb12_img, np.uint8))
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
