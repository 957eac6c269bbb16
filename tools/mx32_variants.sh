#!/bin/bash
# Build variants of libkgwas.so that differ only in score_mx32.hip's compile-time switches (kernel experiments):
#   tools/mx32_variants.sh name1:"-DFLAG=1 ..." name2:"..."   ->  tools/bin/libkgwas_<name>.so
# then time them on the GPU with  KGWAS_MX32=1 KGWAS_LIB=tools/bin/libkgwas_<name>.so python tools/coarse_time.py  (tools/mx_ab.sh)
set -e
cd "$(dirname "$0")/../kmersgwas_amd/csrc"
make -s -j16 >/dev/null
mkdir -p ../../tools/bin
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -I../../include \
      $flags -c score_mx32.hip -o ../../tools/bin/score_mx32_$name.o
  objs=$(ls build/*.o | grep -v score_mx32.o)
  g++ -shared -fPIC $objs ../../tools/bin/score_mx32_$name.o -o ../../tools/bin/libkgwas_$name.so -pthread
  echo "built tools/bin/libkgwas_$name.so ($flags)"
done
