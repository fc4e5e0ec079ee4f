import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from cuttlefish_amd import Context, Format, Type, make_params, synth
img = synth.photo(256, 256, seed=1)
with Context(0) as ctx:
    for f in range(int(Format.ASTC_4x4), int(Format.ASTC_12x12) + 1):
        for q in (2, 3):
            sys.stderr.write("%s q%d: " % (Format(f).name, q)); sys.stderr.flush()
            ctx.encode([img], make_params(Format(f), Type.UNorm, q))
