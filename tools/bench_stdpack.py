#!/usr/bin/env python3
"""Kernel-only throughput of the uncompressed ("standard") converters -- the one HBM-bound
kernel of the path.  Device-resident RGBA32F source (what
StandardConverter reads: 16 B/pixel) -> tightly packed pixels; hipEvent timing via
cfhip_profile_begin/end.  Prints one JSON line per (format, type) with the algorithmic
bandwidth (16 + bytes_per_pixel per pixel) and its fraction of the 8 TB/s HBM3E peak.
usage (GPU box): python tools/bench_stdpack.py [--size 8192] [--steps 20] [--src f32|u8|f16]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--src", default="f32")
    args = ap.parse_args()
    import torch
    from cuttlefish_amd import Context, Format, PixelType, Type, make_params, payload_size

    n = args.size
    if args.src == "f32":
        src = torch.rand((n, n, 4), dtype=torch.float32, device="cuda")*1.2 - 0.1
        ptype, sb = PixelType.RGBA32F, 16
    elif args.src == "f16":
        src = (torch.rand((n, n, 4), dtype=torch.float32, device="cuda")*1.2 - 0.1).half()
        ptype, sb = PixelType.RGBA16F, 8
    else:
        src = torch.randint(0, 256, (n, n, 4), dtype=torch.uint8, device="cuda")
        ptype, sb = PixelType.RGBA8, 4
    ctx = Context(0)
    stream = torch.cuda.current_stream().cuda_stream
    cases = [(Format.R8, Type.UNorm), (Format.R4G4, Type.UNorm), (Format.R5G6B5, Type.UNorm),
             (Format.R8G8B8, Type.UNorm), (Format.R8G8B8A8, Type.UNorm), (Format.B8G8R8A8, Type.UNorm),
             (Format.A2B10G10R10, Type.UNorm), (Format.B10G11R11_UFloat, Type.UFloat),
             (Format.E5B9G9R9_UFloat, Type.UFloat), (Format.R16G16B16, Type.UNorm),
             (Format.R16G16B16A16, Type.Float), (Format.R16G16B16A16, Type.UNorm),
             (Format.R32G32B32, Type.Float), (Format.R32G32B32A32, Type.Float)]
    for fmt, typ in cases:
        nbytes = payload_size(fmt, typ, n, n)
        out = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        surf = [{"pixels": src.data_ptr(), "pixel_type": ptype, "width": n, "height": n,
                 "row_pitch_bytes": n*sb, "out": out.data_ptr(), "out_capacity": nbytes}]
        p = make_params(fmt, typ, 2)
        for _ in range(3):
            ctx.encode_device(surf, p, stream)
        torch.cuda.synchronize()
        ctx.profile_begin()
        for _ in range(args.steps):
            ctx.encode_device(surf, p, stream)
        ms, launches = ctx.profile_end()
        ms /= launches
        algo = n*n*sb + nbytes
        gbs = algo/ms/1e6
        print(json.dumps({"format": fmt.name, "type": typ.name, "src": args.src, "size": n,
                          "bytes_per_pixel": nbytes//(n*n), "kernel_ms": round(ms, 4),
                          "gpix_s": round(n*n/ms/1e6, 2), "algo_gb_s": round(gbs, 1),
                          "hbm_frac": round(gbs/HBM_PEAK_GBS, 3)}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
