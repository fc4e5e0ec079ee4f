#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point cfhip_encode (what HipConverter calls):
upload + kernels + download + sync, pageable numpy buffers.  Also the C5-shaped batch
(texture array with full mip chains).  usage (GPU box): python tools/bench_host.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from cuttlefish_amd import Context, Format, Type, make_params, synth
    ctx = Context(0)
    p = make_params(Format.BC7, Type.UNorm, 2)
    img = synth.photo(4096, 4096, seed=1)
    ctx.encode([img], p)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        ctx.encode([img], p)
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"case": "BC7 Normal 4096x4096 RGBA8, host buffers", "ms": round(dt * 1e3, 2),
                      "mpix_s": round(16.777216 / dt, 1), "kernel_ms": round(ctx.last_kernel_ms(), 2)}))
    # what the Converter adapter holds: RGBAF scanlines, bottom-up (FreeImage) -> negative pitch
    imgf = np.ascontiguousarray((img.astype(np.float32) / 255.0)[::-1])[::-1]
    assert imgf.strides[0] < 0
    ctx.encode([imgf], p)
    t0 = time.perf_counter()
    for _ in range(reps):
        outf = ctx.encode([imgf], p)
    dt = (time.perf_counter() - t0) / reps
    same = bool(np.array_equal(outf[0], ctx.encode([img], p)[0]))
    print(json.dumps({"case": "BC7 Normal 4096x4096 RGBA32F bottom-up host image (quantised by host threads)",
                      "ms": round(dt * 1e3, 2), "mpix_s": round(16.777216 / dt, 1),
                      "kernel_ms": round(ctx.last_kernel_ms(), 2), "same_payload_as_rgba8": same}))
    # C5 shape, scaled: 16 textures x (1024^2 + full mip chain) in ONE call
    base = synth.photo(1024, 1024, seed=2)
    chain = []
    for t in range(16):
        im = np.roll(base, 37 * t, axis=1)
        while True:
            chain.append(np.ascontiguousarray(im))
            if im.shape[0] == 1:
                break
            h = max(im.shape[0] // 2, 1)
            im = im[::2, ::2][:h, :h]
    ctx.encode(chain, p)
    t0 = time.perf_counter()
    ctx.encode(chain, p)
    dt = time.perf_counter() - t0
    px = sum(c.shape[0] * c.shape[1] for c in chain)
    print(json.dumps({"case": "BC7 Normal, 16 x (1024^2 + 11 mips) = %d surfaces in one call" % len(chain),
                      "ms": round(dt * 1e3, 2), "mpix_s": round(px / dt / 1e6, 1),
                      "kernel_ms": round(ctx.last_kernel_ms(), 2)}))
    ctx.close()


if __name__ == "__main__":
    main()
