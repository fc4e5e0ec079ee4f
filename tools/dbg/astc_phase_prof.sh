#!/bin/bash
# Wall-clock share of every phase of the ASTC kernel, from s_memtime stamps in one wave (GPU box):
#   bash tools/dbg/astc_phase_prof.sh [format] [qualities]
# Uses tools/ab/astc_prof.so (a -DCF_ASTC_PROF=1 build, not the product library); builds it when it is not there.
fmt=${1:-ASTC_6x6}; qs=${2:-"0 2 3"}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/tools/ab
if [ ! -f $R/tools/ab/astc_prof.so ]; then
( cd $R/cuttlefish_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-function \
    -DCF_ASTC_PROF=1 -o $R/tools/ab/astc_prof.so astc_encode.hip bc15_encode.hip bc6h_encode.hip bc7_encode.hip cfhip_api.hip etc_encode.hip mipgen.hip std_pack.hip )
fi
for q in $qs; do
  CFHIP_LIB=$R/tools/ab/astc_prof.so python $R/tools/bench_formats.py --size 2048 --steps 1 --formats $fmt --qualities $q 2>&1 | grep "astc prof\|\"format\"" | cut -c1-600
done
