#!/usr/bin/env python3
"""What each step of the BC7 block search buys, measured on the CPU oracle before it is given to a
Texture::Quality level (cfo_bc7_lab_block: every budget field set from here).  Prints RGBA PSNR on the
sampled blocks of tools/quality_tables.py and the gap to the wide search.

    python tools/bc7_lab.py [--blocks 2048] [--content opaque|alpha|both] name=knobs ...

knobs: name:value pairs over the budget fields (defaults = Texture::Quality::Normal), e.g. N4=top:4 H=top:4,sets:3
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
from cuttlefish_amd import synth    # noqa: E402

FIELDS = "iters,m6only,two,mode3,three,rot,n1,n3,n7,n7low,n0,n2,top,starts,uber,uber2,sets,starts3,estq,wide,m4,own,utop".split(",")
NORMAL = dict(iters=1, m6only=0, two=1, mode3=1, three=1, rot=1, n1=6, n3=5, n7=11, n7low=14, n0=5, n2=5,
              top=4, starts=15, uber=0, uber2=1, sets=1, starts3=3, estq=4, wide=0, m4=255, own=0, utop=0)


def knobs(**kw):
    d = dict(NORMAL)
    d.update(kw)
    return [d[f] for f in FIELDS]


def sample_blocks(img, count, rng):
    h, w = img.shape[:2]
    ys = rng.integers(0, h // 4, count) * 4
    xs = rng.integers(0, w // 4, count) * 4
    return np.stack([img[y:y + 4, x:x + 4].reshape(16, -1) for y, x in zip(ys, xs)])


def contents(count, which, kind="photo", group="a"):
    rng = np.random.default_rng(20260929)
    out = []
    for label, alpha in (("opaque", False), ("alpha", True)):
        if which not in ("both", label):
            continue
        if kind == "real":
            import real_lib as R
            if alpha and group != "a":
                continue
            blocks = R.blocks4(count, alpha=alpha, group=group)
            out.append(("real %s %s" % (group, label), np.ascontiguousarray(blocks.reshape(len(blocks), 64))))
            continue
        img = synth.photo(512, 512, seed=21)
        if alpha:
            img[..., 3] = synth.photo(512, 512, seed=22)[..., 0]
        else:
            img[..., 3] = 255
        blocks = sample_blocks(img, count, rng).astype(np.uint8)
        out.append((label, np.ascontiguousarray(blocks.reshape(count, 64))))
    return out


def run(blocks, fn, threads=8):
    """fn(block_ptr, out_ptr) encodes one block; returns total SSE of the decoded blocks."""
    L = O.lib()
    n = len(blocks)
    outs = np.zeros((n, 16), np.uint8)
    decs = np.zeros((n, 64), np.uint8)

    def work(lo, hi):
        for i in range(lo, hi):
            fn(blocks[i].ctypes.data, outs[i].ctypes.data)
            L.cfo_decode_bc7(outs[i].ctypes.data, decs[i].ctypes.data)
    step = (n + threads - 1) // threads
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda k: work(k * step, min(n, (k + 1) * step)), range(threads)))
    d = decs.astype(np.int64) - blocks.astype(np.int64)
    return float((d * d).sum())


def psnr(sse, n):
    return 10.0 * np.log10(255.0 ** 2 * n * 64 / max(sse, 1e-9))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=2048)
    ap.add_argument("--content", default="both")
    ap.add_argument("--no-wide", action="store_true")
    ap.add_argument("--kind", default="photo", help="photo = the synthetic generator, real = tests/golden/real_blocks.npz")
    ap.add_argument("--group", default="a", help="photograph group of the real blocks (b = held out)")
    ap.add_argument("cfg", nargs="*")
    a = ap.parse_args()
    L = O.lib()
    L.cfo_bc7_wide_search.restype = ctypes.c_uint32
    L.cfo_bc7_wide_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(O.Params)]
    L.cfo_encode_bc7_block.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(O.Params)]
    L.cfo_bc7_lab_block.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(O.Params), ctypes.c_void_p]
    L.cfo_decode_bc7.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    cfgs = []
    for c in a.cfg:
        name, _, val = c.partition("=")
        kw = {}
        for item in val.split(","):
            if item:
                k, _, v = item.partition(":")
                kw[k] = int(v)
        cfgs.append((name, kw))
    for label, blocks in contents(a.blocks, a.content, a.kind, a.group):
        n = len(blocks)
        print("== %s, %d blocks" % (label, n))
        cache = "/tmp/bc7_wide_%s_%d.npy" % (label.replace(" ", "_"), n)
        wide = None
        if not a.no_wide:
            if os.path.exists(cache):
                wide = float(np.load(cache))
            else:
                p = O.make_params(36, 0, 4)
                t0 = time.time()
                wide = run(blocks, lambda b, o: L.cfo_bc7_wide_search(b, o, ctypes.byref(p)))
                np.save(cache, wide)
                print("   (wide search: %.0f s)" % (time.time() - t0))
            print("%-28s %.3f dB" % ("wide search", psnr(wide, n)))
        for q in range(5):
            p = O.make_params(36, 0, q)
            t0 = time.time()
            sse = run(blocks, lambda b, o: L.cfo_encode_bc7_block(b, o, ctypes.byref(p)))
            dt = time.time() - t0
            print("%-28s %.3f dB  gap %.3f  (%.2f s)" % ("Q%d" % q, psnr(sse, n),
                                                      psnr(wide, n) - psnr(sse, n) if wide else 0, dt))
        for name, kw in cfgs:
            p = O.make_params(36, 0, 2)
            kn = (ctypes.c_int * 23)(*knobs(**kw))
            t0 = time.time()
            sse = run(blocks, lambda b, o: L.cfo_bc7_lab_block(b, o, ctypes.byref(p), kn))
            dt = time.time() - t0
            print("%-28s %.3f dB  gap %.3f  (%.2f s)  %s" % (name, psnr(sse, n),
                                                          psnr(wide, n) - psnr(sse, n) if wide else 0, dt, kw))


if __name__ == "__main__":
    main()
