"""ASTC HDR GPU-vs-oracle parity probe: mismatching blocks per (footprint, quality, alpha profile)."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from cuttlefish_amd import Alpha, Context, Format, Type, make_params, synth

def img_of(w, h, seed, alpha):
    rng = np.random.default_rng(seed)
    img = synth.hdr_probe(w, h, seed=seed).astype(np.float32)
    if alpha == "ldr":
        img[..., 3] = rng.random((h, w)).astype(np.float32)
    elif alpha == "hdr":
        img[..., 3] = np.exp2(rng.random((h, w))*10.0 - 5.0).astype(np.float32)
    else:
        img[..., 3] = 1.0
    return np.ascontiguousarray(img)

with Context(0) as ctx:
    for fmt in (Format.ASTC_4x4, Format.ASTC_6x6, Format.ASTC_8x6, Format.ASTC_12x12):
        for (al, kind) in ((Alpha.None_, "one"), (Alpha.PreMultiplied, "ldr"), (Alpha.Standard, "hdr")):
            img = img_of(96, 72, int(fmt), kind)
            for q in (0, 2, 3, 4):
                want = O.encode(img, int(fmt), typ=int(Type.UFloat), quality=q, threads=16, alpha=int(al))
                got = ctx.encode([img], make_params(fmt, Type.UFloat, q, alpha=al))[0]
                w = want.reshape(-1, 16); g = got.reshape(-1, 16)
                bad = np.nonzero((w != g).any(axis=1))[0]
                print(fmt.name, kind, q, "blocks", len(w), "mismatch", len(bad), bad[:6].tolist(), flush=True)
                if len(bad) and "-v" in sys.argv:
                    k = bad[0]
                    print("  want", w[k].tobytes().hex(), "\n  got ", g[k].tobytes().hex())
