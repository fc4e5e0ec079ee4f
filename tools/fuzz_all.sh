#!/bin/bash
# Randomised parity sweep over every block format (GPU box): bash tools/fuzz_all.sh [cases] [seed]
cases=${1:-300}; seed=${2:-2}
R=${GRAFT_REPO_ROOT:-$PWD}
for f in BC1_RGB BC1_RGBA BC2 BC3 BC4 BC5 BC6H BC7 ETC1 ETC2_R8G8B8 ETC2_R8G8B8A1 ETC2_R8G8B8A8 EAC_R11 EAC_R11G11 \
         ASTC_4x4 ASTC_5x4 ASTC_5x5 ASTC_6x5 ASTC_6x6 ASTC_8x5 ASTC_8x6 ASTC_8x8 ASTC_10x5 ASTC_10x6 ASTC_10x8 ASTC_10x10 ASTC_12x10 ASTC_12x12; do
  python $R/tools/fuzz_parity.py --cases $cases --seed $seed --format $f 2>&1 | tail -3
done
