#!/bin/bash
# Device-only compile of one .hip file: per-kernel register / scratch / LDS / occupancy report
# (the compiler's own kernel-resource-usage remarks) and the ISA in /tmp/<name>.s.
#   tools/kres.sh cuttlefish_amd/csrc/astc_encode.hip [extra hipcc flags]
src=$1; shift
name=$(basename "$src" .hip)
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S \
  -o /tmp/$name.s "$OLDPWD/$src" -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  grep -E "remark|error" | sed -E 's/^.*(remark|error): +//; s/ \[-Rpass.*$//' |
  awk '/Function Name/{if(line)print line; sub(/Function Name: /,""); line=$0; next}
       /VGPRs:|ScratchSize|Occupancy|LDS Size|SGPRs Spill|VGPRs Spill|TotalSGPRs|error/{gsub(/^ +/,""); line=line" | "$0} END{print line}'
