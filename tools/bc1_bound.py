#!/usr/bin/env python3
"""BC1 / BC3 colour blocks against their TRUE optimum (round-4 VERDICT item 8; GPU box only).

tools/bounds/bc1_optimum.hip walks all 2^32 RGB565 endpoint pairs of a block on the GPU (a tools-only program,
compiled here with hipcc, never part of the library) and returns per block E4 (best four-colour block: what BC2 / BC3
colour and BC1 in c0 > c1 order can reach) and E3 (best three-colour + black block).  This script runs it on the
first N opaque blocks of tests/golden/real_blocks.npz and on N sampled blocks of the synthetic tile, compares the
oracle's BC1_RGB (bound min(E4, E3)) and BC3 (bound E4) at every Texture::Quality, prints the table and writes the
optimum as a fixture (tests/golden/bc1_optimum.npz via gpurun_out/) so that the CPU suite can hold the ladder to it.

    python tools/bc1_bound.py [--blocks 512]
"""
import argparse
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O          # noqa: E402
import real_lib as R            # noqa: E402
from cuttlefish_amd import synth    # noqa: E402


def synth_blocks(count):
    img = synth.photo(512, 512, seed=21)
    img[..., 3] = 255
    rng = np.random.default_rng(20260929)
    ys = rng.integers(0, 128, count) * 4
    xs = rng.integers(0, 128, count) * 4
    return np.ascontiguousarray(np.stack([img[y:y + 4, x:x + 4] for y, x in zip(ys, xs)]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=512)
    a = ap.parse_args()
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    exe = "/tmp/bc1_optimum"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-w", "-o", exe,
                           os.path.join(ROOT, "tools", "bounds", "bc1_optimum.hip")])
    # (real_b: the held-out photograph group of round 6)
    sets = {"real": R.blocks4(a.blocks), "real_b": R.blocks4(a.blocks, group="b"), "synth": synth_blocks(a.blocks)}
    fixture = {}
    print("| blocks | format | Q0 | Q1 | Q2 | Q3 | Q4 | TRUE optimum | gap at Normal | gap at High | gap at Highest |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for name, blocks in sets.items():
        n = len(blocks)
        blocks.reshape(-1).tofile("/tmp/bc1_blocks.bin")
        t0 = time.time()
        subprocess.check_call([exe, "/tmp/bc1_blocks.bin", "/tmp/bc1_opt.bin"])
        opt = np.fromfile("/tmp/bc1_opt.bin", np.uint32).reshape(n, 2).astype(np.float64)
        sys.stderr.write("%s: %d blocks in %.0f s\n" % (name, n, time.time() - t0))
        fixture[name + "_e4"] = opt[:, 0].astype(np.uint32)
        fixture[name + "_e3"] = opt[:, 1].astype(np.uint32)
        strip = R.strip(blocks)

        def psnr(sse):
            return 10 * np.log10(255.0 ** 2 * n * 48 / max(float(sse), 1e-9))
        for fmt, label, bound in ((29, "BC1_RGB", np.minimum(opt[:, 0], opt[:, 1])), (32, "BC3 colour", opt[:, 0])):
            ps = []
            for q in range(5):
                dec = O.decode(O.encode(strip, fmt, quality=q, threads=8), fmt, 4 * n, 4)
                e = ((dec[..., :3].astype(np.int64) - strip[..., :3]) ** 2).reshape(4, n, 4, 3).sum(axis=(0, 2, 3))
                assert (e >= bound).all(), "%s Q%d: a block decodes better than the enumerated optimum" % (label, q)
                ps.append(psnr(e.sum()))
            pb = psnr(bound.sum())
            print("| %s (%d) | %s | %s | %.3f | %.3f | %.3f | %.3f |" % (
                name, n, label, " | ".join("%.3f" % v for v in ps), pb, pb - ps[2], pb - ps[3], pb - ps[4]))
    np.savez_compressed(os.path.join(out_dir, "bc1_optimum.npz"), blocks=np.int32(a.blocks), **fixture)


if __name__ == "__main__":
    main()
