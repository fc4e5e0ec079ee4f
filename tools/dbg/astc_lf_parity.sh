#!/bin/bash
# GPU box: parity of the ASTC kernel with and without the line-fit seed ranking (the oracle's CFO_ASTC_NO_LINEFIT switch)
R=${GRAFT_REPO_ROOT:-$PWD}
echo "== line fit off on both sides"; CFO_ASTC_NO_LINEFIT=1 CFHIP_LIB=$R/tools/ab/nolf.so python $R/tools/dbg/astc_parity.py 43 47 2>&1 | grep -v "^   blk" | awk '{print}' | grep -v " 0/" | head -20
echo "== line fit on"; python $R/tools/dbg/astc_parity.py 43 45 47 2>&1 | grep -v " 0/" | head -40
