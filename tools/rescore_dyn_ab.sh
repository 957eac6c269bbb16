#!/bin/bash
# rescore_kernel: tiles by ticket against round-robin, grid sizes (feed time and kernels of the headline)
for cfg in "1 2048" "0 2048" "1 1280" "1 2560" "0 2048" "1 2048"; do
  set -- $cfg
  echo "== KGWAS_RESCORE_DYN=$1 KGWAS_RESCORE_GRID=$2"
  KGWAS_RESCORE_DYN=$1 KGWAS_RESCORE_GRID=$2 timeout 300 python tools/coarse_time.py 2>&1 | tail -1 | cut -c1-220
done
