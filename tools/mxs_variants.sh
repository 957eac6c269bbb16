#!/bin/bash
# Build variants of libkgwas.so that differ only in score_mxs.hip's compile-time switches (kernel experiments):
#   tools/mxs_variants.sh name1:"-DFLAG=1 ..." name2:"..."   ->  tools/bin/libkgwas_<name>.so
# then time them on the GPU with  KGWAS_LIB=tools/bin/libkgwas_<name>.so python bench.py ...  (tools/mxs_ablate.sh)
set -e
cd "$(dirname "$0")/../kmersgwas_amd/csrc"
make -s -j16 >/dev/null
mkdir -p ../../tools/bin
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -I../../include \
      -DKGWAS_MXS_BENCH_ONLY $flags -c score_mxs.hip -o ../../tools/bin/score_mxs_$name.o
  objs=$(ls build/*.o | grep -v score_mxs.o)
  g++ -shared -fPIC $objs ../../tools/bin/score_mxs_$name.o -o ../../tools/bin/libkgwas_$name.so -pthread
  echo "built tools/bin/libkgwas_$name.so ($flags)"
done
