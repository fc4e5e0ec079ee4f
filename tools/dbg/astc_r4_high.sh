#!/bin/bash
# C3 (ASTC 6x6 High, 4096x4096) with round 4's search on this round's kernel (GPU box): what the restructured
# walks and tables alone are worth at an equal search.  Builds a -DCF_ASTC_R4_HIGH=1 -DCF_ASTC_NO_LINEFIT library
# under tools/ab/ (its payloads differ from the oracle's).
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/tools/ab
( cd $R/cuttlefish_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-function \
    -DCF_ASTC_R4_HIGH=1 -DCF_ASTC_NO_LINEFIT=1 -o $R/tools/ab/astc_r4high.so astc_encode.hip bc15_encode.hip bc6h_encode.hip bc7_encode.hip cfhip_api.hip etc_encode.hip mipgen.hip std_pack.hip 2>/dev/null )
CFHIP_LIB=$R/tools/ab/astc_r4high.so python $R/tools/bench_formats.py --size 4096 --steps 3 --formats ASTC_6x6 --qualities 2,3 2>/dev/null | grep format
