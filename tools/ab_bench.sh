#!/bin/bash
# A/B harness: run bench.py (kernel-only numbers) against every library variant in tools/ab/
# usage (on the GPU box): bash tools/ab_bench.sh [steps]
steps=${1:-5}
for lib in tools/ab/*.so; do
  echo "== $lib"
  CFHIP_LIB=$PWD/$lib timeout 300 python bench.py --steps $steps --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   Mpix/s %.1f  kernel_ms %.3f' % (d['value'], d['roofline']['avg_kernel_ms']))
"
done
