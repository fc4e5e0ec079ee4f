#!/usr/bin/env python3
"""Ranks the ASTC weight-grid configs per footprint and candidate class from a census.

The encoder can afford to LIST 64 configs per class (and keep 24 weight grids' infill tables in
LDS); which ones is a data question, as in astcenc's block-mode percentile tables.  The census
(oracle/astc_encode.c: cfo_astc_census_image) encodes synthetic content -- the bench "photo"
generator with and without its alpha band, smooth gradients, noise, hard-edged shapes -- and, for
every block and class (one partition, dual plane, 2 / 3 / 4 partitions), finds the best of ALL legal
configs by exact error.  Configs are ranked by win count; the top 64 per class go, in that order,
into astc_cfg_rank.h -- written twice with identical content, for the oracle (oracle/) and for the
library (cuttlefish_amd/csrc/), which must build the same lists.

    python tools/astc_rank_configs.py [--size 256]
"""
import argparse
import ctypes
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from cuttlefish_amd import synth  # noqa: E402

FP = [(4, 4), (5, 4), (5, 5), (6, 5), (6, 6), (8, 5), (8, 6), (8, 8), (10, 5), (10, 6), (10, 8),
      (10, 10), (12, 10), (12, 12)]


def content(size):
    """census images: RGBA8, size x size"""
    rng = np.random.default_rng(0xA57C)
    out = []
    for seed in (201, 202, 203):
        out.append(synth.photo(size, size, seed=seed))                      # with the alpha band
        o = synth.photo(size, size, seed=seed + 10)
        o[..., 3] = 255
        out.append(o)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float64)
    g = np.stack([128 + 100*np.sin(xx/37 + yy/71), 128 + 90*np.cos(xx/53 - yy/29),
                  128 + 80*np.sin(xx/23 + yy/41 + 1), 0*xx + 255], -1)
    out.append(g.round().clip(0, 255).astype(np.uint8))                     # smooth
    g2 = g.copy()
    g2[..., 3] = 128 + 120*np.sin(xx/61 + yy/17)
    out.append(g2.round().clip(0, 255).astype(np.uint8))                    # smooth, independent alpha
    n = (g + rng.normal(0, 12, g.shape)).round().clip(0, 255).astype(np.uint8)
    n[..., 3] = 255
    out.append(n)                                                           # textured
    e = np.zeros((size, size, 4), np.uint8)
    e[..., 3] = 255
    for _ in range(size*size//600):
        x, y, w, h = rng.integers(0, size, 4)
        e[y:y + h//8 + 2, x:x + w//8 + 2, :3] = rng.integers(0, 256, 3)
    out.append(e)                                                           # hard edges
    return [np.ascontiguousarray(i) for i in out]


# Real photographs for the census (round 5): pictures that sit in the build container and are NOT the ones the
# quality fixture (tests/golden/real_blocks.npz) samples its blocks from, so the fixture stays held out.
SK = "/opt/conda/lib/python3.9/site-packages/skimage/data/"
REAL_CENSUS = [SK + "motorcycle_right.png", SK + "ihc.png", SK + "retina.jpg", SK + "hubble_deep_field.jpg",
               SK + "color.png", "/opt/conda/lib/python3.9/site-packages/imageio/resources/images/chelsea.png"]


def real_content(size, exclude=()):
    """centre crops (size x size, opaque) of the census photographs; retina / hubble halved first (they are large)"""
    from PIL import Image
    out = []
    for path in REAL_CENSUS:
        if not os.path.exists(path) or any(x in path for x in exclude):
            continue
        im = Image.open(path).convert("RGB")
        if min(im.size) > 4*size:
            im = im.resize((im.size[0]//2, im.size[1]//2), Image.BOX)
        a = np.asarray(im)
        if a.shape[0] < size or a.shape[1] < size:
            continue
        y0, x0 = (a.shape[0] - size)//2, (a.shape[1] - size)//2
        crop = a[y0:y0 + size, x0:x0 + size]
        out.append(np.ascontiguousarray(np.dstack([crop, np.full((size, size), 255, np.uint8)])))
    return out


def real_alpha_content(size):
    """alpha-carrying census content with a TEXTURED alpha (round 6): the colour of one census photograph's centre crop,
    the luma of another's as alpha.  The synthetic content above only ever had smooth alpha (a ramp band, a slow sine), and
    the dual-plane lists it produced for blocks with alpha are all 2 x 2 / 2 x 3 grids -- the alpha-carrying real blocks
    of the quality fixture (another set of photographs) want 4 x 4 x 4 .. 5 levels there and sat 1 dB under the wide
    search at every level."""
    # (the chelsea crop is left out here: that picture is also one of the quality fixture's group-a photographs, and the
    # alpha-carrying fixture blocks are cut from group a -- the five others are sampled for the fixture OUTSIDE this crop)
    crops = real_content(size, exclude=("chelsea",))
    out = []
    for i in range(len(crops)):
        j = (i + 1) % len(crops)
        im = crops[i].copy()
        im[..., 3] = (crops[j][..., :3].astype(np.uint32) @ np.array([54, 183, 19], np.uint32) >> 8).astype(np.uint8)
        out.append(np.ascontiguousarray(im))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=240)
    ap.add_argument("--grids", default="global", choices=["global", "class"],
                    help="global: the 24 grids with the largest share of wins, lists restricted to them; "
                         "class: plain per-class ranking (the builders stop adding grids at 24)")
    ap.add_argument("--out", default="", help="write only this one file (experiments)")
    ap.add_argument("--static-for", default="8x8,10x6,10x8,10x10,12x10",
                    help="footprints that keep the fixed noise-model order (empty lists): on held-out content\n"
                         "(bench tile crops, not in the census) their census lists scored 0.4-1.2 dB lower --\n"
                         "24 grids cover only 55-75 %% of the winning configs of the large footprints")
    ap.add_argument("--keep", default="6x6,12x12",
                    help="footprints whose rows are taken over from the existing oracle/astc_cfg_rank.h (round 3's census,\n"
                         "which saw the 200 best-scored configs per class).  12x12: on held-out images round 3's list beats\n"
                         "both the full census of round 4 (photo -0.1, alpha-carrying -0.7 dB) and the fixed order\n"
                         "(gradients -0.9 dB).  6x6: the full census adds the two-plane 6x5 / 5x6 grids (+0.2 dB on the\n"
                         "photo image) and a wave then waits for the lane with the 60-weight column: 6x6 Normal 2.49 ->\n"
                         "2.73 ms -- BASELINE config 3 is quoted on 6x6, its list stays")
    ap.add_argument("--real", action="store_true", help="add the census photographs (REAL_CENSUS) and synth.photo2 to the content")
    ap.add_argument("--alpha-real", action="store_true", help="add real_alpha_content (photograph colour, another photograph's luma as alpha)")
    ap.add_argument("--alpha-rows-only", action="store_true", help="write only the rows of the alpha-carrying classes; the opaque rows keep what the file holds")
    ap.add_argument("--only", default="", help="footprints to run the census for (others keep their rows), e.g. 6x6,4x4")
    args = ap.parse_args()
    L = O.lib()
    L.cfo_astc_census_image.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    imgs = content(args.size)
    if args.real:
        imgs += real_content(args.size) + [synth.photo2(args.size, args.size, seed=s) for s in (301, 302)]
    if args.alpha_real:
        imgs += real_alpha_content(args.size)
    only = [FP.index(tuple(int(v) for v in x.split("x"))) for x in args.only.split(",") if x]
    table = np.zeros((14, 10, 64), np.uint16)

    def col_rows(v, dual):          # rows of a lane's LDS column a config needs (csrc/astc_tables.h: col_rows)
        N, M = v & 15, (v >> 4) & 15
        return (2 if dual else 1)*(((M*(N + (N & 1))) + 1) & ~1) + 2
    # --alpha-rows-only: the new alpha rows may not ask for a taller lane column than the footprint's lists did -- the
    # column sizes the workgroup for EVERY block of the footprint (12x12: 66 -> 74 rows cost its launches a wave)
    old_max = [0]*14
    if args.alpha_rows_only:
        import re
        rows0 = re.findall(r"\{([0-9, ]+)\},", open(os.path.join(ROOT, "oracle", "astc_cfg_rank.h")).read())
        for fi in range(14):
            old_max[fi] = max([col_rows(int(v), k//2 == 1) for k in range(10) for v in rows0[fi*10 + k].split(",") if int(v)] or [0])

    def census(fi):
        bw, bh = FP[fi]
        counts = np.zeros(10*4096, np.uint32)
        for im in imgs:
            L.cfo_astc_census_image(im.ctypes.data, args.size, args.size, bw, bh, 0, counts.ctypes.data)
        return fi, counts.reshape(10, 4096)

    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        for fi, counts in ex.map(census, only if only else range(14)):
            # the library keeps 24 weight grids' infill tables in LDS per footprint: pick them by
            # their share of the wins (each class normalised, classes weighted by how often their
            # candidates win a block: one partition 3, two partitions 2, dual plane 1, three 1, four 0.5)
            cw = [3.0, 1.0, 2.0, 1.0, 0.5]
            gscore = np.zeros(256)
            codes = np.arange(4096)
            gid = codes & 255                                   # N | M << 4
            for k in range(10):
                tot = counts[k].sum()
                if tot:
                    np.add.at(gscore, gid, cw[k//2]*counts[k]/tot)
            grids = set(int(g) for g in np.argsort(-gscore, kind="stable")[:24] if gscore[g] > 0)
            for k in range(10):
                order = np.argsort(-counts[k].astype(np.int64), kind="stable")
                order = [int(c) for c in order if counts[k][c] > 0 and (args.grids == "class" or (c & 255) in grids) and
                         not (old_max[fi] and (k & 1) and col_rows(int(c), k//2 == 1) > old_max[fi])][:64]
                if args.alpha_rows_only and (k & 1) and len(order) < 64:
                    # a short census row is completed from the row the file held (the table builders would fill it from
                    # the noise-model order, which may bring grids taller than the footprint's column)
                    prev = [int(v) for v in rows0[fi*10 + k].split(",") if int(v)]
                    order = (order + [v for v in prev if v not in order])[:64]
                table[fi, k, :len(order)] = order
                tot = int(counts[k].sum())
                cov = int(counts[k][order].sum()) if order else 0
                print("footprint %dx%d class %d alpha %d: %d blocks, %d configs won at least once, "
                      "listed %d on the 24 chosen grids cover %.1f %%" % (
                          FP[fi][0], FP[fi][1], k//2, k & 1, tot, int((counts[k] > 0).sum()), len(order),
                          100.0*cov/max(tot, 1)), flush=True)
    for name in [x for x in args.static_for.split(",") if x]:
        bw, bh = [int(v) for v in name.split("x")]
        table[FP.index((bw, bh))] = 0
    keep = [x for x in args.keep.split(",") if x]
    if only:
        keep = ["%dx%d" % FP[fi] for fi in range(14) if fi not in only]
        args.static_for = ",".join(x for x in args.static_for.split(",") if x and FP.index(tuple(int(v) for v in x.split("x"))) in only)
    if keep or args.alpha_rows_only:
        import re
        rows = re.findall(r"\{([0-9, ]+)\},", open(os.path.join(ROOT, "oracle", "astc_cfg_rank.h")).read())
        assert len(rows) == 140
        for name in keep:
            bw, bh = [int(v) for v in name.split("x")]
            fi = FP.index((bw, bh))
            for k in range(10):
                table[fi, k] = [int(v) for v in rows[fi*10 + k].split(",")]
        if args.alpha_rows_only:
            def col_rows(v, dual):          # rows of a lane's LDS column this config needs (csrc/astc_tables.h: col_rows)
                N, M = v & 15, (v >> 4) & 15
                return (2 if dual else 1)*(((M*(N + (N & 1))) + 1) & ~1) + 2
            for fi in range(14):
                for k in range(0, 10, 2):
                    table[fi, k] = [int(v) for v in rows[fi*10 + k].split(",")]
    lines = ["/* astc_cfg_rank.h -- GENERATED by tools/astc_rank_configs.py (do not edit): per footprint and",
             " * candidate class x alpha, the weight-grid configs (N | M << 4 | weight range << 8) ranked by how",
             " * often each was the best of ALL legal configs in a census of synthetic content; 0 ends a list",
             " * (an empty list = the fixed noise-model order).  The library and the oracle carry the same data. */",
             "static const unsigned short astc_cfg_rank[14][10][64] = {"]
    for fi in range(14):
        lines.append("\t{ /* %dx%d */" % FP[fi])
        for k in range(10):
            lines.append("\t\t{" + ", ".join(str(int(v)) for v in table[fi, k]) + "},")
        lines.append("\t},")
    lines.append("};")
    txt = "\n".join(lines) + "\n"
    paths = [args.out] if args.out else [os.path.join(ROOT, "oracle", "astc_cfg_rank.h"),
                                         os.path.join(ROOT, "cuttlefish_amd", "csrc", "astc_cfg_rank.h")]
    for path in paths:
        open(path, "w").write(txt)
    print("wrote astc_cfg_rank.h (oracle/ and cuttlefish_amd/csrc/)")


if __name__ == "__main__":
    main()
