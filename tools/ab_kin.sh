#!/bin/bash
# A/B of the kinship accumulation inside ONE gpurun call: tools/ab_kin.sh "VAR=a" "VAR=b" ...
for round in 1 2; do
  for v in "$@"; do
    echo -n "$v  "; env $v KIN_CPU_ROWS=500 python tools/kin_line.py
  done
done
