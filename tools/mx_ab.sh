#!/bin/bash
# time every tools/bin/libkgwas_<name>.so named on the command line with tools/coarse_time.py (one gpurun call)
mkdir -p gpurun_out/mxab
for n in "$@"; do
  KGWAS_LIB=$PWD/tools/bin/libkgwas_$n.so timeout 300 python tools/coarse_time.py 2>&1 | tail -1 | tee -a gpurun_out/mxab/times.txt
done
