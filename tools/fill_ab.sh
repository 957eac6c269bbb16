#!/bin/bash
# chunk-size policy experiments on the headline: KGWAS_FILL (planned share of the key list per chunk) x KGWAS_CAP_BUDGET (records per slot)
for cfg in "0.4 4194304" "0.6 4194304" "0.8 4194304" "0.4 8388608" "0.6 8388608" "0.4 16777216" "0.6 16777216"; do
  set -- $cfg
  echo "== KGWAS_FILL=$1 KGWAS_CAP_BUDGET=$2"
  KGWAS_FILL=$1 KGWAS_CAP_BUDGET=$2 EXPECT=1 STEPS=9 timeout 200 python tools/step_breakdown.py 2>&1 | tail -5 | awk '{print $6, $8, "kernels", $21, "replay", $15, $NF, "chunks"}' | tr '\n' ';'; echo
done
