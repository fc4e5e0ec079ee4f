#!/bin/bash
# Randomised parity sweep over the 14 ASTC footprints (GPU box): bash tools/fuzz_astc.sh [cases] [seed]
cases=${1:-1500}; seed=${2:-11}
R=${GRAFT_REPO_ROOT:-$PWD}
for f in ASTC_4x4 ASTC_5x4 ASTC_5x5 ASTC_6x5 ASTC_6x6 ASTC_8x5 ASTC_8x6 ASTC_8x8 ASTC_10x5 ASTC_10x6 ASTC_10x8 ASTC_10x10 ASTC_12x10 ASTC_12x12; do
  python $R/tools/fuzz_parity.py --cases $cases --seed $seed --format $f 2>&1 | tail -1
done
