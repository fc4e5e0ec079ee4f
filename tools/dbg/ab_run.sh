#!/bin/bash
# A/B timing of library variants in ONE box: bash tools/dbg/ab_run.sh <format> <qualities> lib1.so lib2.so ...
fmt=$1; qs=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2 3; do
  for lib in "$@"; do
    echo "== rep $rep $(basename $lib)"
    CFHIP_LIB=$R/$lib python $R/tools/bench_formats.py --size 2048 --steps 5 --formats $fmt --qualities $qs 2>/dev/null | grep format | python3 -c "
import sys, json
print('  '.join('%s/q%d %.3f' % (d['type'][:2], d['quality'], d['kernel_ms']) for d in map(json.loads, sys.stdin)))"
  done
done
