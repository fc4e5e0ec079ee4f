import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
from cuttlefish_amd import ColorSpace, Context, PixelType
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda", 0)
levels = n.bit_length(); dims = [max(1, n >> k) for k in range(levels)]
ctx = Context(0)
stream = torch.cuda.current_stream().cuda_stream
base = bench.gpu_texture(torch, n, 0, dev)
def gen(use_stream, filt=0):
    c = [torch.zeros((d, d, 4), dtype=torch.float32, device=dev) for d in dims[1:]]
    ctx.generate_mips_device(base.data_ptr(), PixelType.RGBA8, n, n, n * 4, [x.data_ptr() for x in c], color_space=ColorSpace.Linear, filter=filt, stream=stream if use_stream else 0)
    torch.cuda.synchronize()
    return c
a = gen(True); b = gen(True); c = gen(False); d = gen(True, 0x100)
for k in range(len(a)):
    print("level", k + 1, dims[k + 1], "stream/stream", bool(torch.equal(a[k], b[k])), "stream/ctx", bool(torch.equal(a[k], c[k])),
          "max|fi - fallback| %.3g" % float((a[k] - d[k]).abs().max()))
import oracle_lib as O
ref = O.mip_chain(base.cpu().numpy(), 3, filter=0)
print("oracle level 1 equal:", np.array_equal(ref[1], a[0].cpu().numpy()), " level 2:", np.array_equal(ref[2], a[1].cpu().numpy()))
print("oracle vs ctx-stream run level 1:", np.array_equal(ref[1], c[0].cpu().numpy()))
