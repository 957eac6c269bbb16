#!/bin/bash
# A/B of the streamed paths inside ONE gpurun call: tools/ab_ingest.sh "VAR=a" "VAR=b" ...
for v in "$@"; do
  echo -n "$v  "; env $v timeout 300 python tools/ingest_line.py 2>&1 | tail -1
done
