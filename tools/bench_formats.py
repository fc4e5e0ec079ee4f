#!/usr/bin/env python3
"""Kernel-only throughput of every block format (NOT the contract benchmark -- that is
bench.py).  Device-resident synthetic surfaces, hipEvent timing via cfhip_profile_begin/end.
usage (GPU box): python tools/bench_formats.py [--size 2048] [--steps 5]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--formats", default="", help="comma list of Format names (default: all)")
    ap.add_argument("--qualities", default="0,2,4")
    ap.add_argument("--tile", default="photo", choices=["photo", "photo2"],
                    help="LDR tile: synth.photo (SURVEY 8d) or the camera-like synth.photo2 (1024x1024 repeated)")
    args = ap.parse_args()
    import numpy as np
    import torch
    from cuttlefish_amd import Context, Format, PixelType, Type, make_params, payload_size, synth

    n = args.size
    if args.tile == "photo2":
        base = synth.photo2(min(n, 1024), min(n, 1024), seed=1)
        rep = max(1, n // 1024)
        ldr = torch.from_numpy(np.ascontiguousarray(np.tile(base, (rep, rep, 1)))).cuda()
    else:
        ldr = torch.from_numpy(synth.photo(n, n, seed=1)).cuda()
    hdr = torch.from_numpy(synth.hdr_probe(n, n, seed=4).view(np.uint16).astype(np.int32)
                           .astype(np.uint16).view(np.int16)).cuda()
    hdr32 = torch.from_numpy(synth.hdr_probe(n, n, seed=4).astype(np.float32)).cuda()
    ctx = Context(0)
    stream = torch.cuda.current_stream().cuda_stream
    rows = []
    cases = [(Format.BC1_RGB, Type.UNorm), (Format.BC1_RGBA, Type.UNorm), (Format.BC2, Type.UNorm),
             (Format.BC3, Type.UNorm), (Format.BC4, Type.UNorm), (Format.BC4, Type.SNorm),
             (Format.BC5, Type.UNorm), (Format.BC5, Type.SNorm), (Format.BC6H, Type.UFloat),
             (Format.BC6H, Type.Float), (Format.BC7, Type.UNorm),
             (Format.ETC1, Type.UNorm), (Format.ETC2_R8G8B8, Type.UNorm),
             (Format.ETC2_R8G8B8A1, Type.UNorm), (Format.ETC2_R8G8B8A8, Type.UNorm),
             (Format.EAC_R11, Type.UNorm), (Format.EAC_R11G11, Type.SNorm),
             (Format.ASTC_4x4, Type.UNorm), (Format.ASTC_6x6, Type.UNorm),
             (Format.ASTC_8x8, Type.UNorm), (Format.ASTC_12x12, Type.UNorm),
             (Format.ASTC_6x6, Type.UFloat)]
    if args.formats:
        want = set(args.formats.split(","))
        cases = [c for c in cases if c[0].name in want]
    quals = [int(q) for q in args.qualities.split(",")]
    for fmt, typ in cases:
        is_hdr = fmt == Format.BC6H
        astc_hdr = fmt.name.startswith("ASTC") and typ == Type.UFloat      # HDR profile: RGBA32F source
        src = hdr if is_hdr else (hdr32 if astc_hdr else ldr)
        texel = 8 if is_hdr else (16 if astc_hdr else 4)
        nbytes = payload_size(fmt, typ, n, n)
        out = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        surf = [{"pixels": src.data_ptr(), "pixel_type": PixelType.RGBA16F if is_hdr else
                 (PixelType.RGBA32F if astc_hdr else PixelType.RGBA8), "width": n, "height": n,
                 "row_pitch_bytes": n * texel, "out": out.data_ptr(),
                 "out_capacity": nbytes}]
        for q in quals:
            p = make_params(fmt, typ, q)
            ctx.encode_device(surf, p, stream)
            torch.cuda.synchronize()
            ctx.profile_begin()
            for _ in range(args.steps):
                ctx.encode_device(surf, p, stream)
            ms, launches = ctx.profile_end()
            ms /= launches
            algo = n * n * texel + nbytes
            rows.append({"format": fmt.name, "type": typ.name, "quality": q, "tile": args.tile,
                         "kernel_ms": round(ms, 3), "mpix_s": round(n * n / ms / 1e3, 1),
                         "algo_gb_s": round(algo / ms / 1e6, 3)})
            print(json.dumps(rows[-1]), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
