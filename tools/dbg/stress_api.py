#!/usr/bin/env python3
"""Ad-hoc stress of the batched entry point on the GPU box: thousands of surfaces in one call
(the C5 shape: 256 chains of 9 levels down to 1x1), very wide / very tall strips, one huge surface."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from cuttlefish_amd import Context, Format, Type, make_params, synth

rng = np.random.default_rng(3)
with Context(0) as ctx:
    surfs = []
    for t in range(256):
        base = rng.integers(0, 256, (256, 256, 4), dtype=np.uint8)
        for k in range(9):
            d = 256 >> k
            surfs.append(np.ascontiguousarray(base[:d, :d]))
    for fmt in (Format.BC7, Format.BC1_RGB, Format.ETC2_R8G8B8A8, Format.ASTC_6x6):
        t0 = time.time()
        outs = ctx.encode(surfs, make_params(fmt, Type.UNorm, 1))
        dt = time.time() - t0
        bad = 0
        for i in rng.integers(0, len(surfs), 40):
            ref = O.encode(surfs[i], int(fmt), quality=1, threads=4)
            bad += not np.array_equal(np.asarray(outs[i]), ref)
        print("%-14s %d surfaces in one call: %.3f s, %d of 40 sampled payloads differ" % (fmt.name, len(surfs), dt, bad))
    for (w, h) in ((16384, 8), (8, 16384), (1, 1), (3, 5), (8192, 8192)):
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        out = ctx.encode([img], make_params(Format.BC3, Type.UNorm, 2))[0]
        if w*h <= 16384*8:
            ref = O.encode(img, int(Format.BC3), quality=2, threads=8)
            ok = np.array_equal(np.asarray(out), ref)
        else:
            strip = img[:64]
            ok = np.array_equal(np.asarray(out)[:ref_len] if (ref_len := O.encode(strip, int(Format.BC3), quality=2, threads=8).size) else out,
                                O.encode(strip, int(Format.BC3), quality=2, threads=8))
        print("BC3 %dx%d: %d bytes, equals oracle%s: %s" % (w, h, out.size, "" if w*h <= 16384*8 else " (first 16 block rows)", ok))
