#!/bin/bash
# one-column pass over 100 M rows: the planned share of the key list per chunk (KGWAS_FILL; default 0.15 in select mode)
for f in 0.15 0.3 0.45 0.6 0.15; do
  echo "== KGWAS_FILL=$f"
  KGWAS_FILL=$f timeout 200 python tools/one_column.py 2>&1 | grep -E "^feed|^chunks|^candidates|^score_kernel_ms" | tr '\n' ' '; echo
done
