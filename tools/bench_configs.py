#!/usr/bin/env python3
"""The five BASELINE.json configs on one GPU, kernel-only with sources resident in HBM (C1 also
end to end from host memory, it is the reference's "plumbing" case), one JSON line each:
  C1 BC1 512x512 gradient (TextureTest.cpp:53-61)      C2 BC7 Normal 4096x4096 (bench.py's config)
  C3 ASTC 6x6 High ("thorough") 4096x4096              C4 BC6H UFloat Normal 2048x2048 RGBA16F
  C5 one GPU's share of 256 x (2048x2048 + mips) on 8 GPUs = 32 textures: tools/bench_mips.py
usage (GPU box): python tools/bench_configs.py [--steps 10]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import numpy as np
    import torch
    from cuttlefish_amd import Context, Format, PixelType, Quality, Type, make_params, payload_size, synth

    ctx = Context(0)
    stream = torch.cuda.current_stream().cuda_stream

    def resident(name, img, ptype, pb, fmt, typ, quality):
        h, w = img.shape[:2]
        src = torch.from_numpy(img.view(np.int16) if img.dtype == np.float16 else img).cuda()
        nbytes = payload_size(fmt, typ, w, h)
        out = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        surf = [{"pixels": src.data_ptr(), "pixel_type": ptype, "width": w, "height": h,
                 "row_pitch_bytes": w*pb, "out": out.data_ptr(), "out_capacity": nbytes}]
        p = make_params(fmt, typ, quality)
        for _ in range(2):
            ctx.encode_device(surf, p, stream)
        torch.cuda.synchronize()
        ctx.profile_begin()
        for _ in range(args.steps):
            ctx.encode_device(surf, p, stream)
        ms, n = ctx.profile_end()
        ms /= n
        row = {"config": name, "format": fmt.name, "quality": int(quality), "width": w, "height": h,
               "kernel_ms": round(ms, 4), "mpix_s": round(w*h/ms/1e3, 1),
               "algo_gb_s": round((w*h*pb + nbytes)/ms/1e6, 2)}
        print(json.dumps(row), flush=True)
        return row

    grad = synth.gradient(512, 512) if hasattr(synth, "gradient") else None
    if grad is None:
        y, x = np.mgrid[0:512, 0:512].astype(np.float64)
        g = np.stack([x/511, y/511, (511 - x)/511, (511 - y)/511], -1)
        grad = np.round(g*255).astype(np.uint8)
    resident("C1 BC1 512x512 gradient", grad, PixelType.RGBA8, 4, Format.BC1_RGB, Type.UNorm, Quality.Normal)
    p = make_params(Format.BC1_RGB, Type.UNorm, Quality.Normal)
    ctx.encode([grad], p)
    t0 = time.perf_counter()
    for _ in range(50):
        ctx.encode([grad], p)
    dt = (time.perf_counter() - t0)/50
    print(json.dumps({"config": "C1 end to end from host memory (cfhip_encode)", "ms": round(dt*1e3, 4),
                      "mpix_s": round(512*512/dt/1e6, 1)}), flush=True)
    photo = synth.photo(4096, 4096, seed=1)
    resident("C2 BC7 Normal 4096x4096", photo, PixelType.RGBA8, 4, Format.BC7, Type.UNorm, Quality.Normal)
    resident("C3 ASTC 6x6 High 4096x4096", photo, PixelType.RGBA8, 4, Format.ASTC_6x6, Type.UNorm, Quality.High)
    resident("C3' ASTC 6x6 Normal 4096x4096", photo, PixelType.RGBA8, 4, Format.ASTC_6x6, Type.UNorm, Quality.Normal)
    hdr = synth.hdr_probe(2048, 2048, seed=4).astype(np.float16)
    resident("C4 BC6H UFloat Normal 2048x2048 RGBA16F", hdr, PixelType.RGBA16F, 8, Format.BC6H, Type.UFloat, Quality.Normal)
    ctx.close()


if __name__ == "__main__":
    main()
