#!/usr/bin/env python3
"""Mip-chain generation + batched encode of the chain, all resident on the GPU (SURVEY config C5
shape: one 2048x2048 RGBA8 texture with its full 12-level chain, BC7 Normal).  NOT the contract
benchmark (bench.py).  usage (GPU box): python tools/bench_mips.py [--size 2048] [--steps 10]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--textures", type=int, default=0, help="also run a C5-style batch of this many textures")
    args = ap.parse_args()
    import torch
    from cuttlefish_amd import ColorSpace, Context, Format, PixelType, Type, make_params, payload_size, synth

    n = args.size
    levels = n.bit_length()
    img = torch.from_numpy(synth.photo(n, n, seed=1)).cuda()
    dsts = [torch.empty((max(1, n >> k), max(1, n >> k), 4), dtype=torch.float32, device="cuda")
            for k in range(1, levels)]
    ctx = Context(0)
    stream = torch.cuda.current_stream().cuda_stream
    p = make_params(Format.BC7, Type.UNorm, 2)
    outs = [torch.empty(payload_size(Format.BC7, Type.UNorm, max(1, n >> k), max(1, n >> k)),
                        dtype=torch.uint8, device="cuda") for k in range(levels)]
    surf = [{"pixels": img.data_ptr(), "pixel_type": PixelType.RGBA8, "width": n, "height": n,
             "row_pitch_bytes": n*4, "out": outs[0].data_ptr(), "out_capacity": outs[0].numel()}]
    for k, d in enumerate(dsts, start=1):
        w = max(1, n >> k)
        surf.append({"pixels": d.data_ptr(), "pixel_type": PixelType.RGBA32F, "width": w, "height": w,
                     "row_pitch_bytes": w*16, "out": outs[k].data_ptr(), "out_capacity": outs[k].numel()})
    px = sum(max(1, n >> k)**2 for k in range(levels))
    for cs in (ColorSpace.Linear, ColorSpace.sRGB):
        def mips():
            ctx.generate_mips_device(img.data_ptr(), PixelType.RGBA8, n, n, n*4, [d.data_ptr() for d in dsts],
                                     color_space=cs, filter=0, stream=stream)
        mips()
        ctx.encode_device(surf, p, stream)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        for _ in range(args.steps):
            mips()
        e[1].record()
        for _ in range(args.steps):
            ctx.encode_device(surf, p, stream)
        e[2].record()
        torch.cuda.synchronize()
        mip_ms = e[0].elapsed_time(e[1])/args.steps
        enc_ms = e[1].elapsed_time(e[2])/args.steps
        # HBM bytes of the chain: level k reads level k-1 once and writes 16 B/texel
        rd = n*n*4 + sum(max(1, n >> k)**2*16 for k in range(1, levels - 1))
        wr = sum(max(1, n >> k)**2*16 for k in range(1, levels))
        print(json.dumps({"case": "%dx%d RGBA8 + %d-level chain, BC7 Normal, %s" % (n, n, levels, cs.name),
                          "mipgen_ms": round(mip_ms, 3), "mipgen_gb_s": round((rd + wr)/mip_ms/1e6, 1),
                          "encode_chain_ms": round(enc_ms, 3),
                          "chain_mpix_s": round(px/(mip_ms + enc_ms)/1e3, 1),
                          "encode_only_mpix_s": round(px/enc_ms/1e3, 1)}), flush=True)
    # C5 per-GPU share: 256 textures / 8 GPUs = 32 textures of 2048^2 with full chains, mips
    # generated on the GPU, all 32 x 12 = 384 surfaces encoded by ONE batched launch
    if args.textures:
        T = args.textures
        bases = [torch.from_numpy(synth.photo(n, n, seed=100 + t)).cuda() for t in range(min(T, 4))]
        bases = [bases[t % len(bases)].roll(37*t, dims=1).contiguous() for t in range(T)]
        chains = [[torch.empty((max(1, n >> k), max(1, n >> k), 4), dtype=torch.float32, device="cuda")
                   for k in range(1, levels)] for _ in range(T)]
        surf, keep = [], []
        for t in range(T):
            for k in range(levels):
                w = max(1, n >> k)
                o = torch.empty(payload_size(Format.BC7, Type.UNorm, w, w), dtype=torch.uint8, device="cuda")
                keep.append(o)
                src = bases[t] if k == 0 else chains[t][k - 1]
                surf.append({"pixels": src.data_ptr(), "pixel_type": PixelType.RGBA8 if k == 0 else PixelType.RGBA32F,
                             "width": w, "height": w, "row_pitch_bytes": w*(4 if k == 0 else 16),
                             "out": o.data_ptr(), "out_capacity": o.numel()})
        # group by pixel type so that each type is one launch
        surf.sort(key=lambda d: int(d["pixel_type"]))

        def run():
            for t in range(T):
                ctx.generate_mips_device(bases[t].data_ptr(), PixelType.RGBA8, n, n, n*4,
                                         [d.data_ptr() for d in chains[t]], color_space=ColorSpace.sRGB,
                                         filter=0, stream=stream)
            ctx.encode_device(surf, p, stream)
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)/3
        print(json.dumps({"case": "%d textures x (%dx%d + %d-level chain), sRGB Box mips on GPU + BC7 Normal, "
                                  "%d surfaces in 2 launches" % (T, n, n, levels, len(surf)),
                          "ms": round(ms, 2), "mpix_s": round(T*px/ms/1e3, 1)}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
