#!/usr/bin/env python3
"""Quick GPU-vs-oracle parity report for the ASTC kernel (GPU box).  Test infrastructure."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from cuttlefish_amd import Context, Format, Type, make_params, synth

def main():
    fmts = [int(a) for a in sys.argv[1:]] or [43, 47, 50, 56]
    img = synth.photo(96, 72, seed=11)
    img2 = img.copy(); img2[..., 3] = 255
    with Context(0) as ctx:
        for fmt in fmts:
            for name, im in (("alpha", img), ("opaque", img2)):
                for q in range(5):
                    ref = O.encode(im, fmt, quality=q, threads=16).reshape(-1, 16)
                    t = time.time()
                    got = ctx.encode([im], make_params(Format(fmt), Type.UNorm, q))[0].reshape(-1, 16)
                    ms = ctx.last_kernel_ms()
                    bad = np.flatnonzero((ref != got).any(axis=1))
                    print("fmt %d %s q%d: %d/%d blocks differ, kernel %.3f ms" % (fmt, name, q, bad.size, ref.shape[0], ms), flush=True)
                    for i in bad[:3]:
                        print("   blk %d ref %s got %s" % (i, ref[i].tobytes().hex(), got[i].tobytes().hex()))

if __name__ == "__main__":
    main()
