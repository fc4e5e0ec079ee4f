#!/bin/bash
# A/B harness for the standard-format packer: hbm_frac per format for every library in tools/ab/
for lib in tools/ab/std_*.so; do
  echo "== $lib"
  for rep in 1 2; do
  CFHIP_LIB=$PWD/$lib python tools/bench_stdpack.py 2>/dev/null | python -c "
import sys, json
rows = [json.loads(l) for l in sys.stdin if l.startswith('{')]
print(' '.join('%s/%d:%.3f' % (r['format'][:6], r['bytes_per_pixel'], r['hbm_frac']) for r in rows))
"
  done
done
