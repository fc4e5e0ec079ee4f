"""Randomised GPU-vs-oracle sweep of the ASTC HDR profiles: every footprint, alpha profile and quality
level on small images of wild values (smooth, noisy, huge range, specials, masked channels, half and
float sources).  usage (GPU box): python tools/dbg/hdr_fuzz.py [cases] [seed]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from cuttlefish_amd import Alpha, Context, Format, Type, make_params

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fmts = [Format(v) for v in range(int(Format.ASTC_4x4), int(Format.ASTC_12x12) + 1)]
alphas = [Alpha.None_, Alpha.Standard, Alpha.PreMultiplied, Alpha.Encoded]
bad = 0
with Context(0) as ctx:
    for k in range(cases):
        fmt = fmts[int(rng.integers(len(fmts)))]
        al = alphas[int(rng.integers(4))]
        q = int(rng.integers(5))
        w, h = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        kind = int(rng.integers(5))
        if kind == 0:      # smooth ramps
            y, x = np.mgrid[0:h, 0:w]
            img = np.stack([np.exp2((x*rng.random() + y*rng.random())*0.05 + rng.normal()*3) for _ in range(4)], -1)
        elif kind == 1:    # noise over many octaves
            img = np.exp2(rng.normal(size=(h, w, 4))*6)
        elif kind == 2:    # nearly flat
            img = np.full((h, w, 4), 1.0) * np.exp2(rng.normal(size=4)*4) * (1 + rng.normal(size=(h, w, 4))*1e-3)
        elif kind == 3:    # LDR-like
            img = rng.random((h, w, 4))
        else:              # blocks of constants with edges
            img = np.exp2(rng.normal(size=(h//4 + 1, w//4 + 1, 4))*5).repeat(4, 0).repeat(4, 1)[:h, :w]
        img = img.astype(np.float32)
        sp = rng.random((h, w, 4)) < 0.03
        img = np.where(sp, rng.choice(np.array([-1.0, 0.0, np.nan, np.inf, 1e9, 6e-8, 65504.0, 1.0], np.float32), (h, w, 4)), img)
        if rng.random() < 0.3:
            img[..., 3] = 1.0
        if rng.random() < 0.2:             # grey: the luminance modes 2 / 3
            img[..., 1] = img[..., 0]; img[..., 2] = img[..., 0]
        src = img.astype(np.float16) if rng.random() < 0.3 else img
        mask = tuple(bool(rng.random() > 0.15) for _ in range(4))
        src = np.ascontiguousarray(src)
        want = O.encode(src, int(fmt), typ=int(Type.UFloat), quality=q, threads=8, alpha=int(al), mask=tuple(int(m) for m in mask))
        got = ctx.encode([src], make_params(fmt, Type.UFloat, q, alpha=al, color_mask=mask))[0]
        if not np.array_equal(want, got):
            bad += 1
            print("MISMATCH", k, fmt.name, al, q, w, h, kind, src.dtype, mask, flush=True)
print("hdr fuzz: %d cases, %d mismatching" % (cases, bad))
