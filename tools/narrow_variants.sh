#!/bin/bash
# Build variants of libkgwas.so that differ only in score_narrow.hip's compile-time switches (narrow filter experiments):
#   tools/rescore_variants.sh name1:"-DFLAG=1 ..." ...  ->  tools/bin/libkgwas_<name>.so   (time them with tools/mx_ab.sh)
set -e
cd "$(dirname "$0")/../kmersgwas_amd/csrc"
make -s -j16 >/dev/null
mkdir -p ../../tools/bin
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -I../../include $flags -c score_narrow.hip -o ../../tools/bin/score_narrow_$name.o
  objs=$(ls build/*.o | grep -v score_narrow.o)
  g++ -shared -fPIC $objs ../../tools/bin/score_narrow_$name.o -o ../../tools/bin/libkgwas_$name.so -pthread
  echo "built tools/bin/libkgwas_$name.so ($flags)"
done
