#!/bin/bash
# Share of the planar and T / H stages in the ETC2 kernel's time (GPU box): bash tools/dbg/etc_ablate.sh
# Builds -DCF_ETC_ABLATE=n libraries under tools/ab/ (not the product library); their payloads differ.
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/tools/ab
for a in 0 1 2 3; do
  ( cd $R/cuttlefish_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-function \
      -DCF_ETC_ABLATE=$a -o $R/tools/ab/etc_ab$a.so astc_encode.hip bc15_encode.hip bc6h_encode.hip bc7_encode.hip cfhip_api.hip etc_encode.hip mipgen.hip std_pack.hip 2>/dev/null )
  echo "ablate=$a (1: no planar, 2: no T/H)"
  CFHIP_LIB=$R/tools/ab/etc_ab$a.so python $R/tools/bench_formats.py --size 2048 --steps 3 --formats ETC2_R8G8B8 --qualities 2,3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(' ', d['format'], d['quality'], d['kernel_ms'])"
done
