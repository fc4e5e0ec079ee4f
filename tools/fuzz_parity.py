#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep (BC7 by default): random sizes, content classes, alpha
patterns, qualities, colour spaces and masks.  Test infrastructure: uses oracle/ as the checker.
usage (GPU box): python tools/fuzz_parity.py [--cases 60] [--seed 1] [--format BC7]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_image(rng, w, h):
    import numpy as np
    from cuttlefish_amd import synth
    kind = rng.integers(0, 6)
    if kind == 0:
        img = synth.photo(w, h, seed=int(rng.integers(1, 1 << 30)))
    elif kind == 1:
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)                       # noise
    elif kind == 2:
        img = np.zeros((h, w, 4), np.uint8)                                          # two-colour checker
        c = rng.integers(0, 256, (2, 4), dtype=np.uint8)
        m = (np.add.outer(np.arange(h)//rng.integers(1, 5), np.arange(w)//rng.integers(1, 5)) & 1).astype(bool)
        img[m] = c[0]; img[~m] = c[1]
    elif kind == 3:
        img = np.full((h, w, 4), rng.integers(0, 256, 4, dtype=np.uint8), np.uint8)  # flat
    elif kind == 4:
        g = np.linspace(0, 255, w)[None, :, None]*np.ones((h, 1, 4))
        img = (g*rng.random(4)).astype(np.uint8)                                     # ramps
    else:
        img = synth.photo(w, h, seed=int(rng.integers(1, 1 << 30)))
        img = (img.astype(np.int16) + rng.integers(-40, 41, img.shape)).clip(0, 255).astype(np.uint8)
    a = rng.integers(0, 4)
    if a == 0:
        img[..., 3] = 255
    elif a == 1:
        img[..., 3] = np.where(rng.random((h, w)) < 0.5, 255, img[..., 3])
    elif a == 2:
        img[::, ::, 3] = 255
        bx, by = rng.integers(0, max(1, w//4)), rng.integers(0, max(1, h//4))
        img[by*4:by*4 + 4, bx*4:bx*4 + 4, 3] = rng.integers(0, 255)
    return np.ascontiguousarray(img)


def run(format_name, cases, seed, verbose=True):
    """-> number of mismatching cases"""
    import numpy as np
    import oracle_lib as O
    from cuttlefish_amd import Alpha, ColorSpace, Context, Format, Type, make_params
    fmt = Format[format_name]
    rng = np.random.default_rng(seed)
    bad = 0
    with Context(0) as ctx:
        for case in range(cases):
            w, h = int(rng.integers(1, 140)), int(rng.integers(1, 70))
            img = make_image(rng, w, h)
            q = int(rng.integers(0, 5)) if case % 7 else 4
            if q == 4 and w*h > 64*32:
                q = 3
            cs = int(rng.integers(0, 2))
            mask = tuple(int(v) for v in (rng.random(4) < 0.85)) if case % 5 == 0 else (1, 1, 1, 1)
            if not any(mask):
                mask = (1, 1, 1, 1)
            typ = Type.UNorm
            if fmt == Format.BC6H:
                # HDR content: exp2 of the byte image over 18 stops, +-Inf / huge / tiny / negative
                # values sprinkled in; RGBA16F, RGBA32F and RGBA8 sources; UFloat and Float
                typ = Type.UFloat if rng.integers(0, 2) else Type.Float
                f = np.exp2(img.astype(np.float32)/255.0*18.0 - 9.0)
                if typ == Type.Float:
                    f = f*np.where(rng.random(f.shape) < 0.3, -1.0, 1.0).astype(np.float32)
                sp = rng.random(f.shape) < 0.01
                f = np.where(sp, rng.choice(np.array([0.0, 65504.0, 1e9, -1e9, 6e-8, np.inf], np.float32), f.shape), f)
                k = int(rng.integers(0, 3))
                img = f.astype(np.float16) if k == 0 else (f.astype(np.float32) if k == 1 else img)
                cs, mask = 0, (1, 1, 1, 1)
            elif fmt.name.startswith("ASTC") and rng.integers(0, 3) == 0:
                # HDR profiles: float content over 18 stops with special values; alpha type picks
                # ASTCENC_PRF_HDR_RGB_LDR_A (None / PreMultiplied) or ASTCENC_PRF_HDR
                typ = Type.UFloat
                f = np.exp2(img.astype(np.float32)/255.0*18.0 - 9.0)
                sp = rng.random(f.shape) < 0.01
                f = np.where(sp, rng.choice(np.array([0.0, -2.0, 65504.0, 1e9, 6e-8, np.inf], np.float32), f.shape), f)
                alpha_t = int(rng.integers(0, 4))
                if alpha_t in (0, 2):
                    f[..., 3] = img[..., 3].astype(np.float32)/255.0
                img = np.ascontiguousarray(f.astype(np.float32)) if rng.integers(0, 4) else img
                ref = O.encode(img, int(fmt), int(typ), quality=q, threads=16, color_space=cs, mask=mask, alpha=alpha_t)
                got = ctx.encode([img], make_params(fmt, typ, q, color_space=ColorSpace(cs), alpha=Alpha(alpha_t),
                                                    color_mask=tuple(bool(m) for m in mask)))[0]
                if not np.array_equal(ref, got):
                    bad += 1
                    if verbose:
                        print("MISMATCH %s HDR case %d: %dx%d q%d cs%d mask%s alpha%d" % (fmt.name, case, w, h, q, cs, mask, alpha_t))
                continue
            elif fmt in (Format.BC4, Format.BC5, Format.EAC_R11, Format.EAC_R11G11) and rng.integers(0, 2):
                typ = Type.SNorm
                cs = 0
            ref = O.encode(img, int(fmt), int(typ), quality=q, threads=16, color_space=cs, mask=mask)
            got = ctx.encode([img], make_params(fmt, typ, q, color_space=ColorSpace(cs),
                                                color_mask=tuple(bool(m) for m in mask)))[0]
            if not np.array_equal(ref, got):
                bad += 1
                if verbose:
                    print("MISMATCH %s case %d: %dx%d q%d cs%d mask%s" % (fmt.name, case, w, h, q, cs, mask))
                dump = os.environ.get("FUZZ_DUMP")
                if dump and img.dtype == np.uint8:
                    # first differing block: its texels (edge-replicated like the loader), both encodings
                    bw, bh, bs = O.block_geometry(int(fmt)) if hasattr(O, "block_geometry") else (4, 4, ref.size//(((w + 3)//4)*((h + 3)//4)))
                    nbx = (w + bw - 1)//bw
                    r2, g2 = ref.reshape(-1, bs), got.reshape(-1, bs)
                    b = int(np.nonzero((r2 != g2).any(axis=1))[0][0])
                    by_, bx_ = divmod(b, nbx)
                    ys = np.minimum(np.arange(by_*bh, by_*bh + bh), h - 1)
                    xs = np.minimum(np.arange(bx_*bw, bx_*bw + bw), w - 1)
                    blk = img[np.ix_(ys, xs)]
                    with open(dump, "a") as f:
                        f.write("%s %d q%d cs%d mask%s block %d\n  px %s\n  ref %s\n  got %s\n" % (
                            fmt.name, case, q, cs, "".join(str(m) for m in mask), b, blk.tobytes().hex(),
                            r2[b].tobytes().hex(), g2[b].tobytes().hex()))
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--format", default="BC7")
    args = ap.parse_args()
    bad = run(args.format, args.cases, args.seed)
    print("fuzz %s: %d cases, %d mismatching" % (args.format, args.cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
