import sys, os
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
import oracle_lib as O
from cuttlefish_amd import Context, Format, Type, make_params, synth
ctx=Context(0)
def info(b):
    m=0
    while not (int(b[0])>>m)&1: m+=1
    return m,((int(b[0])|(int(b[1])<<8))>>(m+1))&63
for q in (1,2,3):
    img=synth.photo(96,64,seed=10+q)
    ref=O.encode(img,36,quality=q,threads=8).reshape(-1,16)
    got=ctx.encode([img],make_params(Format.BC7,Type.UNorm,q))[0].reshape(-1,16)
    np.save('gpurun_out/dbg_gpu_q%d.npy'%q, got)
    bad=np.flatnonzero((ref!=got).any(axis=1))
    print('q',q,'bad',bad.size,'of',ref.shape[0])
    for i in bad[:10]:
        bx,by=i%24,i//24
        src=img[by*4:by*4+4,bx*4:bx*4+4].astype(int)
        def err(b): return int(((O.decode(b.copy(),36,4,4).astype(int)-src)**2).sum())
        print('  blk',i,'ref',info(ref[i]),err(ref[i]),'gpu',info(got[i]),err(got[i]))
