#!/usr/bin/env python3
"""Quality evidence that does not rest on the oracle alone (CPU, this container):

1. ours (every Texture::Quality level) against two independent encoders -- Pillow's DDS writer and
   Mesa's software texture compression -- on the fixture images of tests/golden/independent_encoders.json;
2. the gap of every quality level to a WIDE search (cfo_bc7_wide_search / cfo_bc6h_wide_search:
   every mode x partition x rotation, least squares and endpoint perturbation iterated to
   convergence on every candidate) on blocks sampled from the synthetic images.

    python tools/quality_tables.py [--blocks 2048] > profiles/r03_quality_tables.md
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O          # noqa: E402
import real_lib as R            # noqa: E402
from cuttlefish_amd import synth    # noqa: E402

NAMES = {29: "BC1", 31: "BC2", 32: "BC3", 33: "BC4", 34: "BC5", 35: "BC6H", 36: "BC7"}


def sample_blocks(img, count, rng):
    h, w = img.shape[:2]
    ys = rng.integers(0, h // 4, count) * 4
    xs = rng.integers(0, w // 4, count) * 4
    return np.stack([img[y:y + 4, x:x + 4].reshape(16, -1) for y, x in zip(ys, xs)])


def psnr_from_sse(sse, n_values):
    return 10.0 * np.log10(255.0 ** 2 * n_values / max(sse, 1e-9))


def bc7_gap(count, rng, threads=None, kind="synth"):
    """-> [(label, [PSNR at Q0..Q4, PSNR of the wide search])] on `count` blocks per content class: sampled from the
    synthetic tile (kind "synth") or the first `count` of the real-photograph blocks (kind "real")."""
    from concurrent.futures import ThreadPoolExecutor
    L = O.lib()
    L.cfo_bc7_wide_search.restype = ctypes.c_uint32
    L.cfo_bc7_wide_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(O.Params)]
    L.cfo_encode_bc7_block.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(O.Params)]
    L.cfo_decode_bc7.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    threads = threads or min(8, os.cpu_count() or 1)
    rows = []
    for label, alpha in (("opaque", False), ("with alpha", True)):
        if kind == "real":
            blocks = R.blocks4(count, alpha=alpha)
            blocks = np.ascontiguousarray(blocks.reshape(len(blocks), 64))
        else:
            img = synth.photo(512, 512, seed=21)
            if alpha:
                img[..., 3] = synth.photo(512, 512, seed=22)[..., 0]      # alpha that varies in every block
            else:
                img[..., 3] = 255
            blocks = np.ascontiguousarray(sample_blocks(img, count, rng).astype(np.uint8).reshape(count, 64))
        count_ = len(blocks)

        def work(lo, hi):
            out = np.zeros(16, np.uint8)
            dec = np.zeros(64, np.uint8)
            sse = np.zeros(6)
            for b in blocks[lo:hi]:
                for q in range(6):
                    p = O.make_params(36, 0, min(q, 4))
                    if q < 5:
                        L.cfo_encode_bc7_block(b.ctypes.data, out.ctypes.data, ctypes.byref(p))
                    else:
                        L.cfo_bc7_wide_search(b.ctypes.data, out.ctypes.data, ctypes.byref(p))
                    L.cfo_decode_bc7(out.ctypes.data, dec.ctypes.data)     # the ctypes calls release the GIL
                    sse[q] += float(((dec.astype(np.int64) - b.astype(np.int64)) ** 2).sum())
            return sse
        step = (count_ + threads - 1) // threads
        with ThreadPoolExecutor(threads) as ex:
            sse = sum(ex.map(lambda k: work(k * step, min(count_, (k + 1) * step)), range(threads)))
        rows.append((label, [psnr_from_sse(s, count_ * 64) for s in sse]))
    return rows


def etc_gap(count, rng, fmt=38, img=None, threads=None, kind="synth"):
    """-> ([PSNR at Q0..Q4], PSNR of the TRUE optimum, modes the optimum uses) of ETC1 (fmt 37) / ETC2 RGB (38) on
    `count` sampled opaque blocks: cfo_etc_true_optimum enumerates every block the format can express."""
    from concurrent.futures import ThreadPoolExecutor
    L = O.lib()
    L.cfo_etc_true_optimum.restype = ctypes.c_uint32
    L.cfo_etc_true_optimum.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    threads = threads or min(8, os.cpu_count() or 1)
    if kind == "real":
        blocks = R.blocks4(count)
        count = len(blocks)
    else:
        if img is None:
            img = synth.photo(512, 512, seed=21)
        img = img.copy()
        img[..., 3] = 255
        blocks = np.ascontiguousarray(sample_blocks(img, count, rng).astype(np.uint8).reshape(count, 4, 4, 4))

    def work(i):
        out = np.zeros(8, np.uint8)
        return L.cfo_etc_true_optimum(blocks[i].ctypes.data, 1 if fmt == 38 else 0, out.ctypes.data)
    with ThreadPoolExecutor(threads) as ex:
        opt = np.array(list(ex.map(work, range(count))), np.float64)
    strip = np.ascontiguousarray(np.concatenate(list(blocks), axis=1))          # 4 x 4*count
    ps = []
    for q in range(5):
        dec = O.decode_etc(O.encode(strip, fmt, quality=q, threads=threads), fmt, 4 * count, 4)
        e = ((dec[..., :3].astype(np.int64) - strip[..., :3]) ** 2).reshape(4, count, 4, 3).sum(axis=(0, 2, 3))
        assert (e >= opt).all(), "a block decodes better than the enumerated optimum"
        ps.append(psnr_from_sse(float(e.sum()), count * 48))
    return ps, psnr_from_sse(float(opt.sum()), count * 48)


def astc_gap(count, rng, bw, bh, alpha=False, threads=None, kind="synth"):
    """-> ([PSNR at Q0..Q4], PSNR of cfo_astc_wide_search) on `count` sampled bw x bh blocks of the photo content:
    every candidate class (one partition, a second plane on every component, every canonical 2 / 3 / 4-partition
    seed), every legal block mode, every endpoint-mode family forced and measured exactly, the best triples
    iterated.  Measured through the decoder (pinned to Mesa's)."""
    from concurrent.futures import ThreadPoolExecutor
    from cuttlefish_amd import Format
    L = O.lib()
    L.cfo_astc_wide_search.restype = ctypes.c_uint64
    L.cfo_astc_wide_search.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    threads = threads or min(8, os.cpu_count() or 1)
    fmt = int(getattr(Format, "ASTC_%dx%d" % (bw, bh)))
    if kind == "real":
        assert not alpha
        blocks = R.blocks(bw, bh, count)
        count = len(blocks)
    else:
        side = 528                                                       # a multiple of 4, 6, 8 and 12
        img = synth.photo(side, side, seed=21)
        img[..., 3] = synth.photo(side, side, seed=22)[..., 0] if alpha else 255
        ys = rng.integers(0, side // bh, count) * bh
        xs = rng.integers(0, side // bw, count) * bw
        blocks = np.ascontiguousarray(np.stack([img[y:y + bh, x:x + bw] for y, x in zip(ys, xs)]))
    outs = np.zeros((count, 16), np.uint8)

    def work(i):
        L.cfo_astc_wide_search(blocks[i].ctypes.data, bw, bh, 0, outs[i].ctypes.data)
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, range(count)))
    strip = np.ascontiguousarray(np.concatenate(list(blocks), axis=1))
    nval = strip.size if alpha else strip[..., :3].size

    def psnr_of(payload):
        dec, outside = O.decode_astc(payload, fmt, bw * count, bh)
        assert outside == 0
        d = dec.astype(np.int64) - strip
        if not alpha:
            d = d[..., :3]
        return 10.0 * np.log10(255.0 ** 2 * nval / max(float((d * d).sum()), 1e-9))
    wide = psnr_of(outs.reshape(-1))
    ps = [psnr_of(O.encode(strip, fmt, quality=q, threads=threads)) for q in range(5)]
    return ps, wide


def bc6h_gap(count, rng):
    L = O.lib()
    L.cfo_bc6h_wide_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(O.Params)]
    L.cfo_encode_bc6h_block.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(O.Params)]
    hdr = synth.hdr_probe(256, 256, seed=4)
    raw = hdr.view(np.uint16)
    blocks = sample_blocks(raw, count, rng)
    payloads = [np.zeros((count, 16), np.uint8) for _ in range(6)]
    for i, blk in enumerate(blocks):
        b = np.ascontiguousarray(blk.reshape(-1).astype(np.uint16))
        for q in range(5):
            p = O.make_params(35, 4, q)
            L.cfo_encode_bc6h_block(b.ctypes.data, payloads[q][i].ctypes.data, ctypes.byref(p))
        p = O.make_params(35, 4, 4)
        L.cfo_bc6h_wide_search(b.ctypes.data, payloads[5][i].ctypes.data, ctypes.byref(p))
    ref = blocks.reshape(count, 4, 4, 4).transpose(0, 1, 2, 3)
    # lay the sampled blocks out as a strip image 4 x (4*count) to reuse the image decoder
    strip = np.concatenate([ref[i] for i in range(count)], axis=1).view(np.float16)
    res = []
    for pl in payloads:
        dec = O.decode_bc6h(pl.reshape(-1), 4 * count, 4, 4)
        res.append(synth.psnr_log(strip[..., :3], dec))
    return res


def independent_table():
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "independent_encoders.json")))
    import importlib.util
    spec = importlib.util.spec_from_file_location("g", os.path.join(ROOT, "tests", "golden", "make_independent_encoders.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    imgs = gen.images()
    lines = []
    for r in gold["rows"]:
        fmt, name = r["format"], r["image"]
        if name not in ("photo3", "photo3_opaque", "gradient", "hdr_probe"):
            continue
        if fmt == 35:
            src = synth.hdr_probe(128, 128, seed=4)
            ours = [synth.psnr_log(src[..., :3], O.decode_bc6h(O.encode(src, 35, 4, quality=q, threads=4), 128, 128, 4)) for q in range(5)]
        else:
            src = imgs[name]
            if fmt == 29:
                src = src.copy()
                src[..., 3] = 255
            ours = [gen.metric(fmt, src, O.decode(O.encode(src, fmt, quality=q, threads=4), fmt, src.shape[1], src.shape[0]))
                    for q in range(5)]
        lines.append("| %s | %s | %s | %s | %s | %+.2f |" % (
            NAMES[fmt], name, " / ".join("%.2f" % v for v in ours),
            "%.2f" % r["pillow_psnr"] if "pillow_psnr" in r else "-", "%.2f" % r["mesa_psnr"],
            ours[2] - max(v for k, v in r.items() if k.endswith("_psnr"))))
    return gold, lines


def bc1_rows():
    """BC1 / BC3 colour against the TRUE optimum of tests/golden/bc1_optimum.npz (tools/bc1_bound.py, GPU brute force)"""
    fx = np.load(os.path.join(ROOT, "tests", "golden", "bc1_optimum.npz"))
    n = int(fx["blocks"])
    img = synth.photo(512, 512, seed=21)
    img[..., 3] = 255
    rng = np.random.default_rng(20260929)
    ys = rng.integers(0, 128, n) * 4
    xs = rng.integers(0, 128, n) * 4
    sets = (("real photographs", "real", R.blocks4(n)), ("synthetic tile", "synth", np.stack([img[y:y + 4, x:x + 4] for y, x in zip(ys, xs)])))
    rows = []
    for label, key, blocks in sets:
        strip = R.strip(np.ascontiguousarray(blocks))
        e4, e3 = fx[key + "_e4"].astype(np.float64), fx[key + "_e3"].astype(np.float64)
        for fmt, name, bound in ((29, "BC1", np.minimum(e4, e3)), (32, "BC3 colour", e4)):
            ps = []
            for q in range(5):
                dec = O.decode(O.encode(strip, fmt, quality=q, threads=8), fmt, 4 * n, 4)
                e = ((dec[..., :3].astype(np.int64) - strip[..., :3]) ** 2).reshape(4, n, 4, 3).sum(axis=(0, 2, 3))
                ps.append(psnr_from_sse(float(e.sum()), n * 48))
            rows.append((name, "%s, %d blocks; bound = the TRUE optimum (all 2^32 endpoint pairs, tools/bounds/bc1_optimum.hip)" % (label, n),
                         ps, psnr_from_sse(float(bound.sum()), n * 48)))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=2048)
    ap.add_argument("--real-blocks", type=int, default=4096, help="blocks of tests/golden/real_blocks.npz per BC7 / ETC row (ASTC: a quarter, at most 768)")
    a = ap.parse_args()
    rng = np.random.default_rng(20260929)
    gold, lines = independent_table()
    print("# Quality against independent encoders, against a wide search / the true optimum (round 5)\n")
    print("Generated by `tools/quality_tables.py` on the CPU oracle (the kernels emit the same bytes).\n")
    print("## Ours (Lowest / Low / Normal / High / Highest) vs Pillow %s and Mesa (%s)\n" % (gold["pillow"], gold["mesa"]))
    print("PSNR in dB over the channels the format stores (BC6H: log-domain PSNR of the halves); last column = ours at "
          "Normal minus the better independent encoder.\n")
    print("| format | image | ours Q0 / Q1 / Q2 / Q3 / Q4 | Pillow | Mesa | margin at Normal |")
    print("|---|---|---|---|---|---|")
    for l in lines:
        print(l)
    hdr = "| format | content | Q0 | Q1 | Q2 | Q3 | Q4 | bound | gap at Normal | gap at High | gap at Highest |\n|---|---|---|---|---|---|---|---|---|---|---|"

    def row(fmt, content, ps, bound):
        print("| %s | %s | %s | %.3f | %.3f | %.3f | %.3f |" % (fmt, content, " | ".join("%.3f" % v for v in ps[:5]), bound,
                                                              bound - ps[2], bound - ps[3], bound - ps[4]))
    print("\n## Gap to the bound on blocks of REAL photographs (tests/golden/real_blocks.npz: 8 photographs, see make_real_blocks.py)\n")
    print("Bounds: BC7 `cfo_bc7_wide_search` (every mode x partition x rotation x index selector, every fit by a steepest descent on "
          "the quantised endpoint grid from four starts -- a solver that is not the encoder's); ETC1 / ETC2 RGB the TRUE optimum of a "
          "block (`cfo_etc_true_optimum`: every expressible block); ASTC `cfo_astc_wide_search` (every candidate class, every canonical "
          "partition seed, every legal block mode, every endpoint-mode family forced and measured exactly, the best triples iterated); "
          "BC1 / BC3 colour the TRUE optimum by GPU brute force.  RGB(A) PSNR of the blocks.\n")
    print(hdr)
    for label, ps in bc7_gap(a.real_blocks, rng, kind="real"):
        row("BC7", "%s, %d blocks" % (label, a.real_blocks if label == "opaque" else min(a.real_blocks, 1024)), ps, ps[5])
    for fmt, name in ((37, "ETC1"), (38, "ETC2 RGB")):
        ps, opt = etc_gap(a.real_blocks, rng, fmt, kind="real")
        row(name, "opaque, %d blocks" % a.real_blocks, ps, opt)
    for bw, bh in ((4, 4), (5, 5), (6, 6), (8, 8), (10, 10), (12, 12)):
        ps, wide = astc_gap(min(768, max(256, a.real_blocks // 4)), rng, bw, bh, kind="real")
        row("ASTC %dx%d" % (bw, bh), "opaque, %d blocks" % min(768, max(256, a.real_blocks // 4)), ps, wide)
    for name, content, ps, bound in bc1_rows():
        if "real" in content:
            row(name, content, ps, bound)
    print("\n## Gap to the bound on the synthetic tile (synth.photo), %d sampled blocks per row\n" % a.blocks)
    print(hdr)
    for label, ps in bc7_gap(a.blocks, rng):
        row("BC7", label, ps, ps[5])
    for fmt, name in ((37, "ETC1"), (38, "ETC2 RGB")):
        ps, opt = etc_gap(max(256, a.blocks // 2), rng, fmt)
        row(name, "opaque", ps, opt)
    for bw, bh in ((4, 4), (6, 6), (8, 8)):
        for alpha in (False, True):
            ps, wide = astc_gap(max(256, a.blocks // 4), rng, bw, bh, alpha)
            row("ASTC %dx%d" % (bw, bh), "with alpha" if alpha else "opaque", ps, wide)
    ps = bc6h_gap(max(256, a.blocks // 2), rng)
    row("BC6H UF16", "HDR probe (log-domain PSNR; bound = `cfo_bc6h_wide_search`)", ps, ps[5])
    for name, content, ps, bound in bc1_rows():
        if "synthetic" in content:
            row(name, content, ps, bound)


if __name__ == "__main__":
    main()
