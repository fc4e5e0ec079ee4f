#!/usr/bin/env python3
"""Gap of every ladder to its bound on blocks of REAL photographs, image by image and on BOTH photograph groups of
tests/golden/real_blocks.npz (group a: the eight pictures the ladders were balanced on in round 5; group b: five
held-out pictures, sampled outside whatever crop a tool ever read -- make_real_blocks.py).  CPU, oracle only.

    python tools/quality_real.py [--blocks4 1024] [--blocks12 256] [--json out.json]  > profiles/r06_quality_real.md

Every row: PSNR of Lowest .. Highest and of the bound, the gaps at Normal / High / Highest; per image the same, and
the WORST image per family (a pooled figure hides a several-fold spread between pictures: round-5 review).
Bounds: BC7 cfo_bc7_wide_search; ETC1 / ETC2 RGB cfo_etc_true_optimum; EAC (ETC2 RGBA8's alpha block)
cfo_eac_true_optimum; ASTC cfo_astc_wide_search; ASTC HDR profiles cfo_astc_wide_search_hdr on real structure under
synthetic radiometry (hdr_blocks), measured on the 16-bit LNS values.
"""
import argparse
import ctypes
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O          # noqa: E402
import real_lib as R            # noqa: E402
from cuttlefish_amd import Format    # noqa: E402

THREADS = min(8, os.cpu_count() or 1)


def _lib():
    L = O.lib()
    L.cfo_bc7_wide_search.restype = ctypes.c_uint32
    L.cfo_bc7_wide_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(O.Params)]
    L.cfo_encode_bc7_block.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(O.Params)]
    L.cfo_decode_bc7.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.cfo_etc_true_optimum.restype = ctypes.c_uint32
    L.cfo_etc_true_optimum.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.cfo_eac_true_optimum.restype = ctypes.c_uint32
    L.cfo_eac_true_optimum.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.cfo_astc_wide_search.restype = ctypes.c_uint64
    L.cfo_astc_wide_search.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return L


def _pmap(fn, n):
    with ThreadPoolExecutor(THREADS) as ex:
        return list(ex.map(fn, range(n)))


def bc7_sse(blocks):
    """(n, 4, 4, 4) -> sse[6, n]: Lowest .. Highest, wide search (RGBA error)"""
    L = _lib()
    flat = np.ascontiguousarray(blocks.reshape(len(blocks), 64))
    n = len(flat)
    sse = np.zeros((6, n))

    def work(i):
        out = np.zeros(16, np.uint8)
        dec = np.zeros(64, np.uint8)
        for q in range(6):
            p = O.make_params(36, 0, min(q, 4))
            if q < 5:
                L.cfo_encode_bc7_block(flat[i].ctypes.data, out.ctypes.data, ctypes.byref(p))
            else:
                L.cfo_bc7_wide_search(flat[i].ctypes.data, out.ctypes.data, ctypes.byref(p))
            L.cfo_decode_bc7(out.ctypes.data, dec.ctypes.data)
            d = dec.astype(np.int64) - flat[i]
            sse[q, i] = float((d * d).sum())
    _pmap(work, n)
    return sse, 64


def etc_sse(blocks, fmt):
    """ETC1 (37) / ETC2 RGB (38): RGB error; bound = the true optimum"""
    L = _lib()
    n = len(blocks)
    opt = np.array(_pmap(lambda i: L.cfo_etc_true_optimum(blocks[i].ctypes.data, 1 if fmt == 38 else 0, np.zeros(8, np.uint8).ctypes.data), n), np.float64)
    strip = R.strip(blocks)
    sse = np.zeros((6, n))
    for q in range(5):
        dec = O.decode_etc(O.encode(strip, fmt, quality=q, threads=THREADS), fmt, 4 * n, 4)
        e = ((dec[..., :3].astype(np.int64) - strip[..., :3]) ** 2).reshape(4, n, 4, 3).sum(axis=(0, 2, 3))
        assert (e >= opt).all(), "a block decodes better than the enumerated optimum"
        sse[q] = e
    sse[5] = opt
    return sse, 48


def etc2_rgba8_sse(blocks):
    """ETC2 RGBA8 (40): RGBA error; bound = the RGB block's true optimum + the alpha block's EAC optimum"""
    L = _lib()
    n = len(blocks)
    opt_rgb = np.array(_pmap(lambda i: L.cfo_etc_true_optimum(blocks[i].ctypes.data, 1, np.zeros(8, np.uint8).ctypes.data), n), np.float64)

    def eac(i):
        v = np.ascontiguousarray(blocks[i][..., 3].reshape(16).astype(np.int32))
        return L.cfo_eac_true_optimum(v.ctypes.data, 0, np.zeros(8, np.uint8).ctypes.data)
    opt_a = np.array(_pmap(eac, n), np.float64)
    strip = R.strip(blocks)
    sse = np.zeros((6, n))
    for q in range(5):
        dec = O.decode_etc(O.encode(strip, 40, quality=q, threads=THREADS), 40, 4 * n, 4)
        d = (dec.astype(np.int64) - strip) ** 2
        e_rgb = d[..., :3].reshape(4, n, 4, 3).sum(axis=(0, 2, 3))
        e_a = d[..., 3].reshape(4, n, 4).sum(axis=(0, 2))
        assert (e_rgb >= opt_rgb).all() and (e_a >= opt_a).all()
        sse[q] = e_rgb + e_a
    sse[5] = opt_rgb + opt_a
    return sse, 64


def astc_sse(blocks, bw, bh, alpha=False):
    L = _lib()
    fmt = int(getattr(Format, "ASTC_%dx%d" % (bw, bh)))
    n = len(blocks)
    outs = np.zeros((n, 16), np.uint8)
    _pmap(lambda i: L.cfo_astc_wide_search(blocks[i].ctypes.data, bw, bh, 0, outs[i].ctypes.data), n)
    strip = R.strip(blocks)
    nch = 4 if alpha else 3

    def sse_of(payload):
        dec, outside = O.decode_astc(payload, fmt, bw * n, bh)
        assert outside == 0
        d = (dec.astype(np.int64) - strip)[..., :nch]
        return (d * d).reshape(bh, n, bw, nch).sum(axis=(0, 2, 3)).astype(np.float64)
    sse = np.zeros((6, n))
    for q in range(5):
        sse[q] = sse_of(O.encode(strip, fmt, quality=q, threads=THREADS))
    sse[5] = sse_of(outs.reshape(-1))
    return sse, bw * bh * nch


_LNS = None


def _lns_table():
    global _LNS
    if _LNS is None:
        L = O.lib()
        L.cfo_astc_lns16.argtypes = [ctypes.c_uint16]
        L.cfo_astc_lns16.restype = ctypes.c_int
        _LNS = np.array([L.cfo_astc_lns16(h) for h in range(65536)], np.int64)
    return _LNS


def hdr_blocks(blocks, seed=5):
    """Real structure, synthetic radiometry (there is no HDR photograph in this image): the 8-bit texels of the
    photograph blocks through an inverse display curve (v / 255) ^ 2.2 and an exposure of 2 ^ e, e drawn per block from
    -6 .. +8 stops -> RGBA16F blocks, alpha 1.0.  What the HDR profiles see is then real edges, textures and gradients
    at radiances from 1e-7 to 256."""
    rng = np.random.default_rng(seed)
    lin = (blocks[..., :3].astype(np.float64) / 255.0) ** 2.2
    e = rng.uniform(-6.0, 8.0, len(blocks))[:, None, None, None]
    out = np.ones(blocks.shape, np.float16)
    out[..., :3] = (lin * np.exp2(e)).astype(np.float16)
    return out


def astc_hdr_sse(hblocks, bw, bh):
    """ASTC HDR profile (Type::UFloat, Alpha::None -> HDR_RGB_LDR_A): squared error on the 16-bit LNS values of the RGB
    halves (the domain the encoder minimises in); bound = cfo_astc_wide_search_hdr"""
    L = _lib()
    L.cfo_astc_wide_search_hdr.restype = ctypes.c_uint64
    L.cfo_astc_wide_search_hdr.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    fmt = int(getattr(Format, "ASTC_%dx%d" % (bw, bh)))
    n = len(hblocks)
    tab = _lns_table()
    lns = tab[hblocks.view(np.uint16)].astype(np.int32)            # (n, bh, bw, 4)
    lns[..., 3] = 255                                              # Alpha::None: an LDR alpha forced to 1.0
    lns = np.ascontiguousarray(lns.reshape(n, bh * bw, 4))
    outs = np.zeros((n, 16), np.uint8)
    _pmap(lambda i: L.cfo_astc_wide_search_hdr(lns[i].ctypes.data, bw, bh, 0, outs[i].ctypes.data), n)
    strip = R.strip(hblocks)
    ref = tab[strip[..., :3].view(np.uint16)]

    def sse_of(payload):
        dec, outside = O.decode_astc_hdr(payload, fmt, bw * n, bh)
        assert outside == 0
        d = tab[np.ascontiguousarray(dec[..., :3]).view(np.uint16)] - ref
        return (d * d).reshape(bh, n, bw, 3).sum(axis=(0, 2, 3)).astype(np.float64)
    sse = np.zeros((6, n))
    for q in range(5):
        sse[q] = sse_of(O.encode(strip, fmt, typ=4, quality=q, threads=THREADS, alpha=0))
    sse[5] = sse_of(outs.reshape(-1))
    return sse, bw * bh * 3


def bc6h_sse(hblocks):
    """BC6H UF16 on the same real-structure / synthetic-radiometry blocks (4x4): squared error of log2(1 + x) of the RGB
    halves (SURVEY 8d's HDR metric, peak log2(1 + 65504)); bound = cfo_bc6h_wide_search"""
    L = _lib()
    L.cfo_bc6h_wide_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(O.Params)]
    L.cfo_encode_bc6h_block.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(O.Params)]
    n = len(hblocks)
    raw = np.ascontiguousarray(hblocks.view(np.uint16).reshape(n, 64))
    payloads = np.zeros((6, n, 16), np.uint8)

    def work(i):
        for q in range(5):
            p = O.make_params(35, 4, q)
            L.cfo_encode_bc6h_block(raw[i].ctypes.data, payloads[q, i].ctypes.data, ctypes.byref(p))
        p = O.make_params(35, 4, 4)
        L.cfo_bc6h_wide_search(raw[i].ctypes.data, payloads[5, i].ctypes.data, ctypes.byref(p))
    _pmap(work, n)
    strip = R.strip(hblocks)
    ref = np.log2(1.0 + strip[..., :3].astype(np.float64))
    sse = np.zeros((6, n))
    for q in range(6):
        dec = O.decode_bc6h(payloads[q].reshape(-1), 4 * n, 4, 4)
        d = np.log2(1.0 + np.abs(dec[..., :3].astype(np.float64))) - ref
        sse[q] = (d * d).reshape(4, n, 4, 3).sum(axis=(0, 2, 3))
    return sse, 48


def psnr_log(s, nvals):
    return 10.0 * np.log10(np.log2(1.0 + 65504.0) ** 2 * nvals / max(float(s), 1e-12))


def psnr_lns(s, nvals):
    return 10.0 * np.log10(65535.0 ** 2 * nvals / max(float(s), 1e-9))


def psnr(s, nvals):
    return 10.0 * np.log10(255.0 ** 2 * nvals / max(float(s), 1e-9))


def rows_of(sse, vals, img, names, lns=False):
    """-> [(label, n, [psnr Q0..Q4, bound])]: pooled first, then per image (lns: PSNR on 16-bit LNS values, peak 65535;
    lns = "log": PSNR of log2(1 + x), peak log2(1 + 65504))"""
    ps = psnr_log if lns == "log" else (psnr_lns if lns else psnr)
    out = [("pooled", sse.shape[1], [ps(sse[q].sum(), vals * sse.shape[1]) for q in range(6)])]
    if img is not None:
        for k, name in enumerate(names):
            m = img == k
            if m.any():
                out.append((name, int(m.sum()), [ps(sse[q][m].sum(), vals * int(m.sum())) for q in range(6)]))
    return out


def family_rows(n4, n12):
    """every (family, group) -> rows; blocks are interleaved over the images, so the image of block i of a prefix
    of the interleaved order is i mod (number of images)"""
    res = []
    for group in ("a", "b"):
        names = R.image_names(group)
        b4 = R.blocks4(n4, group=group)
        img4 = np.arange(len(b4)) % len(names)
        res.append(("BC7 opaque", group, rows_of(*bc7_sse(b4), img4, names)))
        res.append(("ETC1", group, rows_of(*etc_sse(b4, 37), img4, names)))
        res.append(("ETC2 RGB", group, rows_of(*etc_sse(b4, 38), img4, names)))
        for bw, bh in ((4, 4), (5, 5), (6, 6), (8, 8), (10, 10), (12, 12)):
            b12 = R.blocks(bw, bh, n12, group=group)
            img12 = np.arange(len(b12)) % len(names)
            res.append(("ASTC %dx%d" % (bw, bh), group, rows_of(*astc_sse(b12, bw, bh), img12, names)))
    ba = R.blocks4(min(n4, 1024), alpha=True)
    res.append(("BC7 with alpha", "a", rows_of(*bc7_sse(ba), None, None)))
    res.append(("ETC2 RGBA8", "a", rows_of(*etc2_rgba8_sse(ba), None, None)))
    for bw, bh in ((4, 4), (6, 6), (8, 8)):
        res.append(("ASTC %dx%d with alpha" % (bw, bh), "a", rows_of(*astc_sse(R.blocks_alpha(bw, bh, n12), bw, bh, True), None, None)))
    # the HDR profiles: real structure under synthetic radiometry (hdr_blocks), PSNR on the LNS values
    names = R.image_names("a")
    b4 = R.blocks4(min(n4, 1024))
    res.append(("BC6H UF16 (log2 PSNR)", "a", rows_of(*bc6h_sse(hdr_blocks(b4)), np.arange(len(b4)) % len(names), names, lns="log")))
    for bw, bh in ((4, 4), (6, 6), (8, 8)):
        b12 = R.blocks(bw, bh, n12)
        res.append(("ASTC %dx%d HDR (LNS PSNR)" % (bw, bh), "a", rows_of(*astc_hdr_sse(hdr_blocks(b12), bw, bh), np.arange(len(b12)) % len(names), names, lns=True)))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks4", type=int, default=1024)
    ap.add_argument("--blocks12", type=int, default=256)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    res = family_rows(a.blocks4, a.blocks12)
    print("# Gap to the bound on blocks of real photographs, per image, two photograph groups (round 6)\n")
    print("Generated by `tools/quality_real.py --blocks4 %d --blocks12 %d` on the CPU oracle (the kernels emit the same bytes)." % (a.blocks4, a.blocks12))
    print("Group a = the eight photographs the ladders were balanced on in round 5; group b = five HELD-OUT photographs "
          "(hubble_deep_field, ihc, retina, motorcycle_right, color), blocks sampled outside the centre crop the ASTC "
          "config census once read, so no tool ever saw a texel of them.  Gap = bound - level, dB.\n")
    print("## Summary: pooled and worst image, gap at Normal / High / Highest\n")
    print("| format | group a pooled | group a worst image | group b pooled | group b worst image |")
    print("|---|---|---|---|---|")
    summary = {}
    fams = []
    for fam, group, rows in res:
        if fam not in fams:
            fams.append(fam)
        pooled = rows[0][2]
        worst = None
        for name, n, ps in rows[1:]:
            if worst is None or ps[5] - ps[3] > worst[1][5] - worst[1][3]:
                worst = (name, ps)
        summary[(fam, group)] = (pooled, worst)
    g3 = lambda ps: "%.2f / %.2f / %.2f" % (ps[5] - ps[2], ps[5] - ps[3], ps[5] - ps[4])
    for fam in fams:
        cells = []
        for group in ("a", "b"):
            if (fam, group) in summary:
                pooled, worst = summary[(fam, group)]
                cells += [g3(pooled), "%s: %s" % (worst[0], g3(worst[1])) if worst else "-"]
            else:
                cells += ["-", "-"]
        print("| %s | %s |" % (fam, " | ".join(cells)))
    print("\n(worst image = the one with the largest gap at High.)\n")
    for fam, group, rows in res:
        print("\n## %s, group %s\n" % (fam, group))
        print("| image | blocks | Q0 | Q1 | Q2 | Q3 | Q4 | bound | gap Normal | gap High | gap Highest |")
        print("|---|---|---|---|---|---|---|---|---|---|---|")
        for name, n, ps in rows:
            print("| %s | %d | %s | %.3f | %.3f | %.3f | %.3f |" % (name, n, " | ".join("%.3f" % v for v in ps[:5]), ps[5],
                                                                 ps[5] - ps[2], ps[5] - ps[3], ps[5] - ps[4]))
    if a.json:
        js = {}
        for (fam, group), (pooled, worst) in summary.items():
            js["%s/%s" % (fam, group)] = {"pooled_gap_nhh": [round(pooled[5] - pooled[q], 4) for q in (2, 3, 4)],
                                          "worst_image": worst[0] if worst else None,
                                          "worst_gap_nhh": [round(worst[1][5] - worst[1][q], 4) for q in (2, 3, 4)] if worst else None}
        json.dump(js, open(a.json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
