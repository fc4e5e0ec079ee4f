#!/bin/bash
# Build library variants of ONE kernel file for tools/ab_bench.sh: the other objects come from the product build.
# usage: bash tools/ab_build.sh <file.hip> name1="-DFOO=1" name2="-DBAR=2" ...
set -e
src=$1; shift
base=$(basename $src .hip)
mkdir -p tools/ab
others=$(ls cuttlefish_amd/build/*.o | grep -v "/$base.o")
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function $flags -c -o /tmp/ab_$name.o $src &
done
wait
for spec in "$@"; do
  name=${spec%%=*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/$name.so /tmp/ab_$name.o $others
done
ls -la tools/ab
