#!/bin/bash
# tools/bin/libkgwas_exp.so: the library with its tuning / ablation switches compiled in (csrc/env.h, exp_*), built in a scratch
# copy of the tree; use it with KGWAS_LIB=$PWD/tools/bin/libkgwas_exp.so.
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
B=/tmp/kgwas_exp_build
rm -rf $B/kmersgwas_amd/csrc.new; mkdir -p $B/kmersgwas_amd/csrc $B/include $R/tools/bin
for f in $R/kmersgwas_amd/csrc/*.cpp $R/kmersgwas_amd/csrc/*.h $R/kmersgwas_amd/csrc/*.hip $R/kmersgwas_amd/csrc/Makefile; do
  cmp -s $f $B/kmersgwas_amd/csrc/$(basename $f) || cp $f $B/kmersgwas_amd/csrc/
done
cp $R/include/*.h $B/include/
make -s -C $B/kmersgwas_amd/csrc -j16 EXPERIMENTS=1 ../lib/libkgwas.so 2>&1 | grep -E "error" -A3 || true
cp $B/kmersgwas_amd/lib/libkgwas.so $R/tools/bin/libkgwas_exp.so
echo "built tools/bin/libkgwas_exp.so"
