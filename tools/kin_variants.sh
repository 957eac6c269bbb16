#!/bin/bash
# Build variants of libkgwas.so that differ only in kin_kernels.hip's compile-time switches: tools/kin_variants.sh name:"-DFLAG=1" ...
set -e
cd "$(dirname "$0")/../kmersgwas_amd/csrc"
make -s -j16 >/dev/null
mkdir -p ../../tools/bin
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -I../../include $flags -c kin_kernels.hip -o ../../tools/bin/kin_kernels_$name.o
  objs=$(ls build/*.o | grep -v kin_kernels.o)
  g++ -shared -fPIC $objs ../../tools/bin/kin_kernels_$name.o -o ../../tools/bin/libkgwas_$name.so -pthread
  echo "built tools/bin/libkgwas_$name.so ($flags)"
done
