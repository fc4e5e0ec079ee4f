#!/bin/bash
# Build ablated variants of the library (phases of the ASTC kernel switched off; output is then
# wrong on purpose) and time them back to back on the GPU box:  bash tools/dbg/astc_ablate.sh
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
for lib in $R/tools/ab/astc_abl_*.so; do
  echo "== $(basename $lib)"
  CFHIP_LIB=$lib python $R/tools/bench_formats.py --size 2048 --steps 2 --formats ${1:-ASTC_6x6} --qualities ${2:-3} 2>/dev/null | grep format
done
