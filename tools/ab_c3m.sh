#!/bin/bash
# configs[3] per-GPU shape (2048 samples x 201 columns): mixed operand sets (block-scaled two-slice ramp + int8 one-slice
# steady state, the default) against the int8 sets alone, one gpurun call
mkdir -p gpurun_out/c3
for v in ${C3_VARIANTS:-"A=1" "KGWAS_COARSE_MIXED=0" "A=1" "KGWAS_COARSE_MIXED=0"}; do
  env $v timeout 600 python bench.py --samples 2048 --perms 200 --rows ${C3_ROWS:-100000000} --steps 5 --warmup 2 --no-cpu-baseline --no-subrecords > gpurun_out/c3/line.json 2> gpurun_out/c3/err.txt
  python - "$v" gpurun_out/c3/line.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
h=j["host"]; r=j["roofline"]
print("%-24s step %.2f replay busiest %.2f | kernels %.2f coarse %.2f sets %s" % (sys.argv[1], j["ms_per_step"], h["replay_ms_per_step"], r["all_scoring_kernels_ms_per_step"], r["kernel_ms_per_step"], [(c["slices"], c["tiles_per_lds_group"], c["lds_groups"], c["rows_per_step"], round(c["ms_per_step"],1)) for c in r["coarse_sets"]]))
PY
done
