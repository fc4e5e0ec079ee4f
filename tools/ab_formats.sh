#!/bin/bash
# A/B of library variants on one box through tools/bench_formats.py: bash tools/ab_formats.sh "<bench_formats args>"
for lib in tools/ab/*.so; do
  echo "== $lib"
  CFHIP_LIB=$PWD/$lib python tools/bench_formats.py $1 2>/dev/null | grep format | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   %-16s q%d %.3f ms  %.0f Mpix/s' % (d['format'], d['quality'], d['kernel_ms'], d['mpix_s']))
"
done
