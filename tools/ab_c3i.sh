#!/bin/bash
# configs[3] per-GPU shape (2048 x 201, int8 filter) and configs[1] with the int8 filter for libraries named on the command line
mkdir -p gpurun_out/c3
for spec in "$@"; do
  lib=${spec%%,*}; envs=${spec#*,}; [ "$envs" = "$spec" ] && envs="A=1"
  for shape in "--samples 2048 --perms 200 --rows 100000000" "--rows 100000000"; do
    env KGWAS_LIB=$PWD/tools/bin/libkgwas_$lib.so KGWAS_COARSE_MX=0 $envs timeout 600 python bench.py $shape --steps 5 --warmup 2 --no-cpu-baseline --no-subrecords > gpurun_out/c3/line.json 2> gpurun_out/c3/err.txt
    python - "$spec $shape" gpurun_out/c3/line.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
h=j["host"]; r=j["roofline"]
print("%-60s step %.2f | kernels %.2f coarse %.2f frac %.3f sets %s" % (sys.argv[1][:60], j["ms_per_step"], r["all_scoring_kernels_ms_per_step"], r["kernel_ms_per_step"], r["frac"], [(c["tiles_per_lds_group"], c["lds_groups"], round(c["ms_per_step"],1)) for c in r["coarse_sets"]]))
PY
  done
done
