#!/bin/bash
# 12-wave (168-VGPR) build against the 8-wave (256-VGPR) build of the ASTC kernel, same box; optional extra libraries
R=${GRAFT_REPO_ROOT:-$PWD}
for lib in cuttlefish_amd/libcuttlefish_hip.so "$@"; do
for e in 0 1; do
  if [ $e = 1 ]; then export CFHIP_ASTC_NO_DENSE=1; else unset CFHIP_ASTC_NO_DENSE; fi
  CFHIP_LIB=$R/$lib python $R/tools/bench_formats.py --size 2048 --steps 3 --formats ASTC_6x6,ASTC_4x4,ASTC_8x8,ASTC_12x12 --qualities 0,2,3,4 2>&1 | grep "format\|rror" | grep -v UFloat | python3 -c "
import sys, json
out = []
for l in sys.stdin:
    try:
        d = json.loads(l); out.append('%s/q%d %.3f' % (d['format'][5:], d['quality'], d['kernel_ms']))
    except Exception:
        out.append(l.strip()[:80])
print('$(basename $lib) NO_DENSE=$e', '  '.join(out))"
done
done
