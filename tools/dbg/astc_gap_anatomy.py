#!/usr/bin/env python3
"""Where the ASTC ladder's distance to the wide search sits: per block the decoded error of a level and of
cfo_astc_wide_search, and what each chose (partitions, planes, weight grid, weight range) -- CPU, oracle only.

    python tools/dbg/astc_gap_anatomy.py [--fp 6x6] [--q 3] [--blocks 768] [--image name]
"""
import argparse
import collections
import ctypes
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O          # noqa: E402
import real_lib as R            # noqa: E402
from cuttlefish_amd import Format    # noqa: E402

WRANGE = {(0, 2): 2, (0, 3): 3, (0, 4): 4, (0, 5): 5, (0, 6): 6, (0, 7): 8, (1, 2): 10, (1, 3): 12, (1, 4): 16, (1, 5): 20, (1, 6): 24, (1, 7): 32}


def block_info(blk):
    """-> (partitions, dual, W, H, weight levels) of one 16-byte LDR block (2-D modes); None for void extent"""
    b = int(blk[0]) | (int(blk[1]) << 8)
    if (b & 0x1FF) == 0x1FC:
        return None
    part = ((b >> 11) & 3) + 1
    if b & 3:
        Rr = ((b >> 4) & 1) | ((b & 3) << 1)
        A, B = (b >> 5) & 3, (b >> 7) & 3
        k = (b >> 2) & 3
        if k == 0: W, H = B + 4, A + 2
        elif k == 1: W, H = B + 8, A + 2
        elif k == 2: W, H = A + 2, B + 8
        elif (b >> 8) & 1: W, H = (B & 1) + 2, A + 2
        else: W, H = A + 2, (B & 1) + 6
        D, Hp = (b >> 10) & 1, (b >> 9) & 1
    else:
        Rr = ((b >> 4) & 1) | (((b >> 2) & 3) << 1)
        A, B = (b >> 5) & 3, (b >> 9) & 3
        k = (b >> 7) & 3
        D, Hp = (b >> 10) & 1, (b >> 9) & 1
        if k == 0: W, H = 12, A + 2
        elif k == 1: W, H = A + 2, 12
        elif k == 2: W, H, D, Hp = A + 6, B + 6, 0, 0
        elif A == 0: W, H = 6, 10
        else: W, H = 10, 6
    return part, D, W, H, WRANGE.get((Hp, Rr), 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fp", default="6x6")
    ap.add_argument("--q", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=768)
    ap.add_argument("--image", default=None)
    ap.add_argument("--group", default="a")
    a = ap.parse_args()
    bw, bh = [int(v) for v in a.fp.split("x")]
    fmt = int(getattr(Format, "ASTC_%dx%d" % (bw, bh)))
    L = O.lib()
    L.cfo_astc_wide_search.restype = ctypes.c_uint64
    L.cfo_astc_wide_search.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    blocks = R.blocks(bw, bh, a.blocks, image=a.image, group=a.group)
    n = len(blocks)
    strip = R.strip(blocks)
    wide = np.zeros((n, 16), np.uint8)
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(lambda i: L.cfo_astc_wide_search(blocks[i].ctypes.data, bw, bh, 0, wide[i].ctypes.data), range(n)))
    ours = O.encode(strip, fmt, quality=a.q, threads=8).reshape(n, 16)

    def sse(payload):
        dec, _ = O.decode_astc(payload.reshape(-1), fmt, bw * n, bh)
        d = dec.astype(np.int64)[..., :3] - strip[..., :3]
        return (d * d).reshape(bh, n, bw, 3).sum(axis=(0, 2, 3)).astype(np.float64)
    ew, eo = sse(wide), sse(ours)
    ps = lambda e: 10 * np.log10(255.0 ** 2 * n * bw * bh * 3 / e.sum())
    print("%s Q%d: ours %.3f dB, wide %.3f dB, gap %.3f (%d blocks%s)" % (a.fp, a.q, ps(eo), ps(ew), ps(ew) - ps(eo), n, ", " + a.image if a.image else ""))
    excess = eo - ew
    tot = excess.sum()
    cat = collections.Counter()
    catn = collections.Counter()
    for i in range(n):
        io, iw = block_info(ours[i]), block_info(wide[i])
        if io is None or iw is None:
            key = "void"
        else:
            d = []
            if io[0] != iw[0]: d.append("P%d->%d" % (io[0], iw[0]))
            if io[1] != iw[1]: d.append("dual%d->%d" % (io[1], iw[1]))
            if io[2:4] != iw[2:4]: d.append("grid")
            if io[4] != iw[4]: d.append("range")
            key = "+".join(d) if d else "same structure"
        cat[key] += excess[i]
        catn[key] += 1
    print("share of the squared-error excess by what the wide search chose differently:")
    for k, v in sorted(cat.items(), key=lambda kv: -kv[1]):
        print("  %-36s %6.1f %%  (%d blocks)" % (k, 100.0 * v / tot, catn[k]))
    # by structure of the wide search's block
    gw = collections.Counter(); gn = collections.Counter()
    for i in range(n):
        iw = block_info(wide[i])
        key = "void" if iw is None else "P%d dual%d" % (iw[0], iw[1])
        gw[key] += excess[i]; gn[key] += 1
    print("share by the wide search's partition count / planes:")
    for k, v in sorted(gw.items(), key=lambda kv: -kv[1]):
        print("  %-36s %6.1f %%  (%d blocks)" % (k, 100.0 * v / tot, gn[k]))


if __name__ == "__main__":
    main()
