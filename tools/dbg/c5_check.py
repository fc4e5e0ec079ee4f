"""Debug: batched C5 encode vs per-texture encode, level by level."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from cuttlefish_amd import ColorSpace, Context, Format, PixelType, Type, make_params, payload_size

n, T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda", 0)
levels = n.bit_length()
dims = [max(1, n >> k) for k in range(levels)]
nbytes = [payload_size(Format.BC7, Type.UNorm, d, d) for d in dims]
chain_bytes = sum(nbytes)
ctx = Context(0)
stream = torch.cuda.current_stream().cuda_stream
params = make_params(Format.BC7, Type.UNorm, 2)
bases = [bench.gpu_texture(torch, n, t, dev) for t in range(T)]
chains = [[torch.empty((d, d, 4), dtype=torch.float32, device=dev) for d in dims[1:]] for _ in range(T)]
out = torch.zeros(T * chain_bytes, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()

def surfaces(base_list, chain_list, out_buf):
    s = []
    for i in range(len(base_list)):
        off = i * chain_bytes
        for k, d in enumerate(dims):
            src = base_list[i] if k == 0 else chain_list[i][k - 1]
            s.append({"pixels": src.data_ptr(), "pixel_type": PixelType.RGBA8 if k == 0 else PixelType.RGBA32F,
                      "width": d, "height": d, "row_pitch_bytes": d * (4 if k == 0 else 16),
                      "out": out_buf.data_ptr() + off, "out_capacity": nbytes[k]})
            off += nbytes[k]
    s.sort(key=lambda e: int(e["pixel_type"]))
    return s

for rep in range(2):
    for i in range(T):
        ctx.generate_mips_device(bases[i].data_ptr(), PixelType.RGBA8, n, n, n * 4, [c.data_ptr() for c in chains[i]],
                                 color_space=ColorSpace.Linear, filter=0, stream=stream)
    ctx.encode_device(surfaces(bases, chains, out), params, stream)
torch.cuda.synchronize()
for t in (0, 1, T - 1):
    b = [bench.gpu_texture(torch, n, t, dev)]
    c = [[torch.empty((d, d, 4), dtype=torch.float32, device=dev) for d in dims[1:]]]
    o = torch.zeros(chain_bytes, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.generate_mips_device(b[0].data_ptr(), PixelType.RGBA8, n, n, n * 4, [x.data_ptr() for x in c[0]],
                             color_space=ColorSpace.Linear, filter=0, stream=stream)
    ctx.encode_device(surfaces(b, c, o), params, stream)
    torch.cuda.synchronize()
    print("texture", t, "base equal", bool(torch.equal(b[0], bases[t])))
    off = 0
    for k, d in enumerate(dims):
        same = bool(torch.equal(o[off:off + nbytes[k]], out[t * chain_bytes + off:t * chain_bytes + off + nbytes[k]]))
        mip_same = True if k == 0 else bool(torch.equal(c[0][k - 1], chains[t][k - 1]))
        if not same or not mip_same:
            diff = int((o[off:off + nbytes[k]] != out[t * chain_bytes + off:t * chain_bytes + off + nbytes[k]]).sum())
            print("  level", k, d, "payload equal", same, "(%d bytes differ)" % diff, "mip equal", mip_same)
        off += nbytes[k]
ctx.close()
