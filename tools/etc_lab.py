#!/usr/bin/env python3
"""What each step of the ETC RGB block search buys, measured on the CPU oracle against the TRUE optimum of a block
(cfo_etc_true_optimum) before a Texture::Quality level gets it (cfo_etc_lab_block: every budget field set from here).

    python tools/etc_lab.py [--blocks 4096] [--kind real|photo] [--etc1] name=knob:value,... ...

knobs: walk, radius, refine, nlists, e0, e1, e2 (list ends), cube, lsq, flips (both_flips), joint
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
import oracle_lib as O          # noqa: E402
import real_lib as R            # noqa: E402
from cuttlefish_amd import synth    # noqa: E402

FIELDS = "walk,radius,refine,nlists,e0,e1,e2,cube,lsq,flips,joint,rec,fst,gate".split(",")
BASE = dict(walk=2, radius=2, refine=1, nlists=0, e0=0, e1=0, e2=0, cube=0, lsq=0, flips=0, joint=0, rec=0, fst=0, gate=0)


def blocks_of(kind, count, group="a", image=None):
    if kind == "real":
        return R.blocks4(count, group=group, image=image)
    img = synth.photo(512, 512, seed=21)
    img[..., 3] = 255
    rng = np.random.default_rng(20260929)
    ys = rng.integers(0, 128, count) * 4
    xs = rng.integers(0, 128, count) * 4
    return np.ascontiguousarray(np.stack([img[y:y + 4, x:x + 4] for y, x in zip(ys, xs)]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=4096)
    ap.add_argument("--kind", default="real")
    ap.add_argument("--etc1", action="store_true")
    ap.add_argument("--group", default="a", help="photograph group of tests/golden/real_blocks.npz (b = the held-out one)")
    ap.add_argument("--image", default=None)
    ap.add_argument("--attr", action="store_true", help="attribute the excess error to (mode of the optimum, mode chosen)")
    ap.add_argument("cfg", nargs="*")
    a = ap.parse_args()
    L = O.lib()
    L.cfo_etc_true_optimum.restype = ctypes.c_uint32
    L.cfo_etc_true_optimum.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.cfo_etc_lab_block.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    blocks = blocks_of(a.kind, a.blocks, a.group, a.image)
    n = len(blocks)
    etc2 = 0 if a.etc1 else 1
    fmt = 37 if a.etc1 else 38
    cache = "/tmp/etc_opt_%s%s%s_%d_%d.npy" % (a.kind, a.group, a.image or "", n, etc2)
    if os.path.exists(cache):
        opt = np.load(cache)
    else:
        def work(i):
            out = np.zeros(8, np.uint8)
            return L.cfo_etc_true_optimum(blocks[i].ctypes.data, etc2, out.ctypes.data)
        with ThreadPoolExecutor(8) as ex:
            opt = np.array(list(ex.map(work, range(n))), np.float64)
        np.save(cache, opt)
    strip = R.strip(blocks)

    def sx3(v):
        return v - 8 if v >= 4 else v

    def mode(b):
        b = [int(x) for x in b]
        if not (b[3] & 2):
            return "indiv"
        for k, name in ((0, "T"), (1, "H"), (2, "planar")):
            v = (b[k] >> 3) + sx3(b[k] & 7)
            if v < 0 or v > 31:
                return name
        return "diff"
    optblk = None
    if a.attr:
        cb = "/tmp/etc_optblk_%s%s%s_%d_%d.npy" % (a.kind, a.group, a.image or "", n, etc2)
        if os.path.exists(cb):
            optblk = np.load(cb)
        else:
            optblk = np.zeros((n, 8), np.uint8)

            def workb(i):
                L.cfo_etc_true_optimum(blocks[i].ctypes.data, etc2, optblk[i].ctypes.data)
            with ThreadPoolExecutor(8) as ex:
                list(ex.map(workb, range(n)))
            np.save(cb, optblk)
        omode = [mode(b) for b in optblk]

    def gap(payload):
        dec = O.decode_etc(payload, fmt, 4 * n, 4)
        e = ((dec[..., :3].astype(np.int64) - strip[..., :3]) ** 2).reshape(4, n, 4, 3).sum(axis=(0, 2, 3))
        assert (e >= opt).all()
        if a.attr:
            import collections
            acc = collections.defaultdict(lambda: [0, 0.0])
            m = [mode(b) for b in payload.reshape(-1, 8)]
            flipdiff = 0.0
            for i in range(n):
                acc[(omode[i], m[i])][0] += 1
                acc[(omode[i], m[i])][1] += e[i] - opt[i]
                if omode[i] in ("diff", "indiv") and m[i] in ("diff", "indiv") and ((optblk[i][3] ^ payload.reshape(-1, 8)[i][3]) & 1):
                    flipdiff += e[i] - opt[i]
            tot = (e - opt).sum()
            for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:8]:
                print("        opt=%-6s ours=%-6s n=%4d excess share %.1f%%" % (k[0], k[1], v[0], 100 * v[1] / tot))
            print("        base-colour blocks whose flip differs from the optimum's: %.1f%% of the excess" % (100 * flipdiff / tot))
        return 10 * np.log10(e.sum() / opt.sum()), 10 * np.log10(255.0 ** 2 * n * 48 / e.sum())
    print("optimum %.3f dB  (%d blocks, %s)" % (10 * np.log10(255.0 ** 2 * n * 48 / opt.sum()), n, a.kind))
    for q in range(5):
        g, p = gap(O.encode(strip, fmt, quality=q, threads=8))
        print("%-24s %.3f dB  gap %.3f" % ("Q%d" % q, p, g))
    for c in a.cfg:
        name, _, val = c.partition("=")
        d = dict(BASE)
        for item in val.split(","):
            if item:
                k, _, v = item.partition(":")
                d[k] = int(v)
        kn = (ctypes.c_int * 14)(*[d[f] for f in FIELDS])
        outs = np.zeros((n, 8), np.uint8)

        def work(lo, hi):
            for i in range(lo, hi):
                L.cfo_etc_lab_block(blocks[i].ctypes.data, etc2, kn, outs[i].ctypes.data)
        step = (n + 7) // 8
        with ThreadPoolExecutor(8) as ex:
            list(ex.map(lambda k: work(k * step, min(n, (k + 1) * step)), range(8)))
        g, p = gap(outs.reshape(-1))
        print("%-24s %.3f dB  gap %.3f   %s" % (name, p, g, val))


if __name__ == "__main__":
    main()
