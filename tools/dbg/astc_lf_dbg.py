#!/usr/bin/env python3
"""GPU box: the line-fit keys of one 4x4 block, kernel (tools/ab/lfdbg.so) against oracle.  Test infrastructure."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from cuttlefish_amd import Context, Format, Type, make_params, synth
img = synth.photo(96, 72, seed=11)
blk = np.ascontiguousarray(img[0:4, 0:4])
q = int(sys.argv[1]) if len(sys.argv) > 1 else 3
print(blk.reshape(16, 4).tolist())
sys.stdout.flush()
ref = O.encode(blk, 43, quality=q, threads=1)
sys.stdout.flush()
with Context(0) as ctx:
    got = ctx.encode([blk], make_params(Format(43), Type.UNorm, q))[0]
print("ref", ref.tobytes().hex(), "got", got.tobytes().hex())
