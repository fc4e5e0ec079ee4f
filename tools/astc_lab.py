#!/usr/bin/env python3
"""What each step of the ASTC (LDR) block search buys, measured on the CPU oracle against the wide search
(cfo_astc_wide_search) on blocks of real photographs, before a Texture::Quality level gets it
(cfo_astc_lab_block: the ladder fields and the refinement budget set from here).

    python tools/astc_lab.py [--fp 6x6] [--blocks 256] [--kind real|photo] name=knob:value,... ...

knobs: q (structure: 2 / 3 one pass of the half-wave layout, 4 passes of 8), K (0 = the level's own allocation),
       limit, j2, j3, j4, nd, it (hits iterated after the walk), rounds (of those), maxpass, rall (rounds of every lane)
"""
import argparse
import ctypes
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O          # noqa: E402
import real_lib as R            # noqa: E402
from cuttlefish_amd import Format, synth    # noqa: E402

FIELDS = "q,K,limit,j2,j3,j4,nd,it,rounds,maxpass,rall,xo".split(",")
LADDER = {2: dict(q=2, K=0, limit=64, j2=4, j3=2, j4=0, nd=2), 3: dict(q=3, K=0, limit=256, j2=4, j3=2, j4=0, nd=2),
          4: dict(q=4, K=0, limit=256, j2=14, j3=9, j4=6, nd=2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fp", default="6x6")
    ap.add_argument("--blocks", type=int, default=256)
    ap.add_argument("--kind", default="real")
    ap.add_argument("cfg", nargs="*")
    a = ap.parse_args()
    bw, bh = [int(v) for v in a.fp.split("x")]
    fmt = int(getattr(Format, "ASTC_%dx%d" % (bw, bh)))
    L = O.lib()
    L.cfo_astc_wide_search.restype = ctypes.c_uint64
    L.cfo_astc_wide_search.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.cfo_astc_lab_block.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    if a.kind == "real":
        blocks = R.blocks(bw, bh, a.blocks)
    else:
        side = 528
        img = synth.photo(side, side, seed=21)
        img[..., 3] = 255
        rng = np.random.default_rng(20260929)
        ys = rng.integers(0, side // bh, a.blocks) * bh
        xs = rng.integers(0, side // bw, a.blocks) * bw
        blocks = np.ascontiguousarray(np.stack([img[y:y + bh, x:x + bw] for y, x in zip(ys, xs)]))
    n = len(blocks)
    strip = R.strip(blocks)

    def psnr_of(payload):
        dec, outside = O.decode_astc(payload, fmt, bw * n, bh)
        d = dec.astype(np.int64)[..., :3] - strip[..., :3]
        return 10.0 * np.log10(255.0 ** 2 * d.size / max(float((d * d).sum()), 1e-9))
    cache = "/tmp/astc_wide_%s_%s_%d.npy" % (a.kind, a.fp, n)
    if os.path.exists(cache):
        wide_pl = np.load(cache)
    else:
        wide_pl = np.zeros((n, 16), np.uint8)
        t0 = time.time()
        with ThreadPoolExecutor(8) as ex:
            list(ex.map(lambda i: L.cfo_astc_wide_search(blocks[i].ctypes.data, bw, bh, 0, wide_pl[i].ctypes.data), range(n)))
        np.save(cache, wide_pl)
        print("(wide search: %.0f s)" % (time.time() - t0))
    wide = psnr_of(wide_pl.reshape(-1))
    print("wide search  %.3f dB  (%d blocks %s, %s)" % (wide, n, a.fp, a.kind))
    for q in range(5):
        t0 = time.time()
        p = psnr_of(O.encode(strip, fmt, quality=q, threads=8))
        print("%-22s %.3f dB  gap %.3f  (%.2f s)" % ("Q%d" % q, p, wide - p, time.time() - t0))
    for c in a.cfg:
        name, _, val = c.partition("=")
        kw = {}
        for item in val.split(","):
            if item:
                k, _, v = item.partition(":")
                kw[k] = int(v)
        d = dict(it=0, rounds=0, maxpass=0, rall=0, xo=0)
        d.update(LADDER[kw.get("q", 3)])
        d.update(kw)
        kn = (ctypes.c_int * 12)(*[d[f] for f in FIELDS])
        outs = np.zeros((n, 16), np.uint8)
        t0 = time.time()
        with ThreadPoolExecutor(8) as ex:
            list(ex.map(lambda i: L.cfo_astc_lab_block(blocks[i].ctypes.data, bw, bh, 0, kn, outs[i].ctypes.data), range(n)))
        p = psnr_of(outs.reshape(-1))
        print("%-22s %.3f dB  gap %.3f  (%.2f s)  %s" % (name, p, wide - p, time.time() - t0, val))


if __name__ == "__main__":
    main()
