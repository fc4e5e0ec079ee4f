#!/usr/bin/env python3
"""Linear vs sRGB colour space, kernel-only, Normal, 2048x2048: sRGB images select the
perceptual / REC709 channel weights (S3tcConverter.cpp:196-199, EtcConverter.cpp:60-88), which
run the weighted distance paths of the kernels.  usage (GPU box): python tools/bench_srgb.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cuttlefish_amd import ColorSpace, Context, Format, PixelType, Type, make_params, payload_size, synth

n = 2048
img = torch.from_numpy(synth.photo(n, n, seed=1)).cuda()
ctx = Context(0)
for fmt in (Format.BC1_RGB, Format.BC1_RGBA, Format.BC3, Format.BC7, Format.ETC2_R8G8B8, Format.ETC2_R8G8B8A1,
            Format.ETC2_R8G8B8A8, Format.ASTC_6x6):
    out = torch.empty(payload_size(fmt, Type.UNorm, n, n), dtype=torch.uint8, device="cuda")
    surf = [{"pixels": img.data_ptr(), "pixel_type": PixelType.RGBA8, "width": n, "height": n,
             "row_pitch_bytes": n*4, "out": out.data_ptr(), "out_capacity": out.numel()}]
    row = {"format": fmt.name}
    for cs in (ColorSpace.Linear, ColorSpace.sRGB):
        p = make_params(fmt, Type.UNorm, 2, color_space=cs)
        ctx.encode_device(surf, p)
        torch.cuda.synchronize()
        ctx.profile_begin()
        for _ in range(5):
            ctx.encode_device(surf, p)
        ms, k = ctx.profile_end()
        row[cs.name + "_ms"] = round(ms/k, 3)
        row[cs.name + "_mpix_s"] = round(n*n/(ms/k)/1e3, 1)
    print(json.dumps(row), flush=True)
