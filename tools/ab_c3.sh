#!/bin/bash
# configs[3] per-GPU shape (2048 samples x 201 columns) with the block-scaled and the int8 filter, one gpurun call
mkdir -p gpurun_out/c3
for v in "KGWAS_COARSE_MX=1" "KGWAS_COARSE_MX=0" "KGWAS_COARSE_MX=1" "KGWAS_COARSE_MX=0"; do
  env $v timeout 600 python bench.py --samples 2048 --perms 200 --rows ${C3_ROWS:-100000000} --steps 5 --warmup 2 --no-cpu-baseline --no-subrecords > gpurun_out/c3/line.json 2> gpurun_out/c3/err.txt
  python - "$v" gpurun_out/c3/line.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
h=j["host"]; r=j["roofline"]
print("%-20s step %.2f replay busiest %.2f cpu %.1f | kernels %.2f coarse %.2f frac %.3f sets %s" % (sys.argv[1], j["ms_per_step"], h["replay_ms_per_step"], h["replay_cpu_ms_per_step"], r["all_scoring_kernels_ms_per_step"], r["kernel_ms_per_step"], r["frac"], [(c["tiles_per_lds_group"], c["lds_groups"], round(c["ms_per_step"],1)) for c in r["coarse_sets"]]))
PY
done
