#!/bin/bash
# A/B of environment-switchable variants inside ONE gpurun call (boxes and co-tenants differ between calls):
#   tools/ab_env.sh "VAR=a VAR2=b" "VAR=c" ...   -> one bench line summary per variant, two rounds interleaved
mkdir -p gpurun_out/ab
for round in 1 2; do
  i=0
  for v in "$@"; do
    i=$((i+1))
    env $v timeout 200 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-subrecords $AB_BENCH_ARGS > gpurun_out/ab/v${i}_r${round}.json 2> gpurun_out/ab/v${i}_r${round}.err
    python - "$v" gpurun_out/ab/v${i}_r${round}.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
h=j["host"]
print("%-40s step %.2f (median %.2f) replay_cpu %.1f busiest %.2f tail %.2f gpu_kernels %.2f coarse %.2f" % (sys.argv[1], j["ms_per_step"], h["step_ms_median"], h["replay_cpu_ms_per_step"], h["replay_ms_per_step"], h["replay_tail_ms_per_step"], j["roofline"]["all_scoring_kernels_ms_per_step"], j["roofline"]["kernel_ms_per_step"]))
PY
  done
done
