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
Round 6 -- a SECOND group of photographs, held out from every ladder (no tuning tool reads it; it exists to show that
the gaps measured on the first group generalise), arrays with the suffix _b:
    images_b   hubble_deep_field, ihc, retina, motorcycle_right, color (skimage sample data)
    rgb4_b / rgb4_b_img, rgb12_b / rgb12_b_img    as above, 256 / 64 per image
    The ASTC config census (tools/astc_rank_configs.py --real, the 5x5 / 6x6 lists) read the 240 x 240 CENTRE crop
    of these five pictures (retina and hubble halved first); the blocks here are sampled OUTSIDE that crop, so no
    texel of this group was seen by any tool.  BC7 and ETC never had a census.
    rgba12     (NA12, 12, 12, 4)  12x12 patches of a first-group photograph with another first-group photograph's
                                  luma as alpha (smooth + textured alpha), for the alpha-carrying ASTC rows
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
PHOTOS_B = [
    ("hubble_deep_field", SK + "hubble_deep_field.jpg", 2),      # (name, path, the census's reduction factor)
    ("ihc", SK + "ihc.png", 1),
    ("retina", SK + "retina.jpg", 2),
    ("motorcycle_right", SK + "motorcycle_right.png", 1),
    ("color", SK + "color.png", 1),
]
CENSUS_CROP = 240                   # tools/astc_rank_configs.py --size
N4_PER_IMAGE_B = 256
N12_PER_IMAGE_B = 64
NA12 = 256
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
    # ---- the second, held-out group (its own generator: the first group's arrays stay what they were) ----
    rngb = np.random.default_rng(0xB10C5)
    rgb4_b, rgb4_b_img, rgb12_b, rgb12_b_img = [], [], [], []
    for k, (_, path, red) in enumerate(PHOTOS_B):
        img = load(path, "RGB")
        h, w = img.shape[:2]
        c = CENSUS_CROP * red
        cy0, cx0 = (h - c) // 2, (w - c) // 2

        def outside(y, x, s):       # the s x s patch at (y, x) shares no texel with the census's centre crop
            return y + s <= cy0 or y >= cy0 + c or x + s <= cx0 or x >= cx0 + c
        for size, count, dst, dsti in ((4, N4_PER_IMAGE_B, rgb4_b, rgb4_b_img), (12, N12_PER_IMAGE_B, rgb12_b, rgb12_b_img)):
            got = 0
            while got < count:
                y, x = int(rngb.integers(0, h // size)) * size, int(rngb.integers(0, w // size)) * size
                if not outside(y, x, size):
                    continue
                dst.append(img[y:y + size, x:x + size])
                dsti.append(k)
                got += 1
    rgba12 = []
    while len(rgba12) < NA12:
        a, b = rngb.choice(len(photos), 2, replace=False)
        ia, ib = photos[a], photos[b]
        h, w = min(ia.shape[0], ib.shape[0]), min(ia.shape[1], ib.shape[1])
        y, x = int(rngb.integers(0, h // 12)) * 12, int(rngb.integers(0, w // 12)) * 12
        lum = (ib[y:y + 12, x:x + 12].astype(np.uint32) @ np.array([54, 183, 19], np.uint32) >> 8).astype(np.uint8)
        rgba12.append(np.dstack([ia[y:y + 12, x:x + 12], lum]))
    out = os.path.join(HERE, "real_blocks.npz")
    np.savez_compressed(
        out, images=np.array([n for n, _ in PHOTOS]),
        rgb4=np.stack(rgb4).astype(np.uint8), rgb4_img=np.array(rgb4_img, np.uint8),
        rgba4=np.stack(rgba4).astype(np.uint8),
        rgb12=np.stack(rgb12).astype(np.uint8), rgb12_img=np.array(rgb12_img, np.uint8),
        images_b=np.array([n for n, _, _ in PHOTOS_B]),
        rgb4_b=np.stack(rgb4_b).astype(np.uint8), rgb4_b_img=np.array(rgb4_b_img, np.uint8),
        rgb12_b=np.stack(rgb12_b).astype(np.uint8), rgb12_b_img=np.array(rgb12_b_img, np.uint8),
        rgba12=np.stack(rgba12).astype(np.uint8))
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
