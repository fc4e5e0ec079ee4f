import sys, json
sys.path.insert(0, '/root/repo')
import torch
from cuttlefish_amd import Context, Format, PixelType, Type, ColorSpace, make_params, payload_size, synth
n = 4096
img = torch.from_numpy(synth.photo(n, n, seed=1)).cuda()
ctx = Context(0)
out = torch.empty(payload_size(Format.BC7, Type.UNorm, n, n), dtype=torch.uint8, device="cuda")
surf = [{"pixels": img.data_ptr(), "pixel_type": PixelType.RGBA8, "width": n, "height": n, "row_pitch_bytes": n*4, "out": out.data_ptr(), "out_capacity": out.numel()}]
for cs in (ColorSpace.Linear, ColorSpace.sRGB):
    p = make_params(Format.BC7, Type.UNorm, 2, color_space=cs)
    ctx.encode_device(surf, p); torch.cuda.synchronize()
    ctx.profile_begin()
    for _ in range(3): ctx.encode_device(surf, p)
    ms, k = ctx.profile_end()
    print(cs.name, round(ms/k, 3), "ms", round(n*n/(ms/k)/1e3, 1), "Mpix/s")
