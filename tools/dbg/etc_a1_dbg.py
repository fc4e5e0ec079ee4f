import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle_lib as O
from cuttlefish_amd import Context, Format, Type, ColorSpace, make_params
cases=[("1e764cff1e764cff616223ff616223ff1e764cff1e764cff616223ff616223ff1e764cff1e764cff616223ff616223ff1e764cff1e764cff616223ff616223ff",4,1),
("a7fc1dff547b62ffd53752ff9830a8ffa7fc1dff547b62ffd53752ff9830a8ffa7fc1dff547b62ffd53752ff9830a8ffa7fc1dff547b62ffd53752ff9830a8ff",4,1),
("7e07dc9a7e07dc9a7e07dc9a7e07dc9a38cfbf4038cfbf4038cfbf4038cfbf40be93e787be93e787be93e787be93e78762570c7762570c7762570c7762570c77",1,4)]
with Context(0) as ctx:
    for px,w,h in cases:
        p=np.ascontiguousarray(np.frombuffer(bytes.fromhex(px),np.uint8).reshape(4,4,4)[:h,:w].copy())
        for fmt in (Format.ETC2_R8G8B8A1, Format.ETC2_R8G8B8, Format.ETC2_R8G8B8A8, Format.ETC1):
            for cs in (0,1):
                row=[]
                for q in range(5):
                    ref=O.encode(p,int(fmt),quality=q,threads=1,color_space=cs)
                    got=ctx.encode([p],make_params(fmt,Type.UNorm,q,color_space=ColorSpace(cs)))[0]
                    row.append("ok" if np.array_equal(ref,got) else "DIFF %s/%s"%(ref.tobytes().hex(),got.tobytes().hex()))
                print(w,h,fmt.name, "cs",cs, row)
