#!/bin/bash
# 12-wave (168-VGPR) build against the 8-wave (256-VGPR) build of the ASTC kernel, same box
R=${GRAFT_REPO_ROOT:-$PWD}
for e in 0 1; do
  if [ $e = 1 ]; then export CFHIP_ASTC_NO_DENSE=1; fi
  python $R/tools/bench_formats.py --size 2048 --steps 3 --formats ASTC_6x6,ASTC_4x4 --qualities 3,4 2>/dev/null | grep format | python3 -c "
import sys, json
print('NO_DENSE=$e', '  '.join('%s/%s/q%d %.3f' % (d['format'], d['type'][:2], d['quality'], d['kernel_ms']) for d in map(json.loads, sys.stdin)))"
done
