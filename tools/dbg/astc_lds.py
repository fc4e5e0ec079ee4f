"""Print the ASTC launch plan (waves per workgroup, dynamic LDS) per footprint and quality."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
os.environ["CFHIP_ASTC_DEBUG"] = "1"
import numpy as np
from cuttlefish_amd import Context, Format, Type, make_params, synth
img = synth.photo(96, 96, seed=1)
with Context(0) as ctx:
    for fmt in (Format.ASTC_4x4, Format.ASTC_5x5, Format.ASTC_6x5, Format.ASTC_6x6, Format.ASTC_8x6, Format.ASTC_8x8, Format.ASTC_10x10, Format.ASTC_12x12):
        for q in (0, 3, 4):
            sys.stderr.write("%s q%d: " % (fmt.name, q)); sys.stderr.flush()
            ctx.encode([img], make_params(fmt, Type.UNorm, q))
