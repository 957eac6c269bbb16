#!/bin/bash
# Build variants of libkgwas.so that differ only in ONE kernel file's compile-time switches (kernel experiments):
#   tools/variants.sh score_narrow name1:"-DFLAG=1 ..." name2:"..."   ->  tools/bin/libkgwas_<name>.so
# then run anything with  KGWAS_LIB=tools/bin/libkgwas_<name>.so
set -e
file=$1; shift
cd "$(dirname "$0")/../kmersgwas_amd/csrc"
make -s -j16 >/dev/null
mkdir -p ../../tools/bin
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -I../../include \
      $flags -c $file.hip -o ../../tools/bin/${file}_$name.o
  objs=$(ls build/*.o | grep -v "build/$file.o")
  g++ -shared -fPIC $objs ../../tools/bin/${file}_$name.o -o ../../tools/bin/libkgwas_$name.so -pthread
  echo "built tools/bin/libkgwas_$name.so ($flags)"
done
