#!/bin/bash
# A/B of the one-column scan inside ONE gpurun call: tools/ab_p1.sh "VAR=a" "VAR=b" ...
mkdir -p gpurun_out/ab
for round in 1 2; do
  i=0
  for v in "$@"; do
    i=$((i+1))
    env $v timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ingest > gpurun_out/ab/p1_v${i}_r${round}.json 2> gpurun_out/ab/p1_v${i}_r${round}.err
    echo -n "$v  "; python tools/p1_line.py gpurun_out/ab/p1_v${i}_r${round}.json
  done
done
