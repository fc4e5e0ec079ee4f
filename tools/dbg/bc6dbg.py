import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import oracle_lib as O
from cuttlefish_amd import Context, Format, Type, make_params, synth
img = synth.hdr_probe(64, 64, seed=3).astype(np.float16)
with Context(0) as ctx:
    for q in (0,1,2):
        got = ctx.encode([img], make_params(Format.BC6H, Type.UFloat, q))[0].reshape(-1,16)
        ref = O.encode(img, int(Format.BC6H), int(Type.UFloat), quality=q, threads=4).reshape(-1,16)
        bad = np.where((got!=ref).any(axis=1))[0]
        print("q",q,"bad blocks",len(bad),"of",len(got))
        for b in bad[:4]:
            g=int.from_bytes(got[b].tobytes(),'little'); r=int.from_bytes(ref[b].tobytes(),'little')
            print("  blk",b,"got mode bits",bin(g&31),"ref",bin(r&31), hex(g), hex(r))
