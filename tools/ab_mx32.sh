#!/bin/bash
# A/B of the two forms of the block-scaled filter on ONE box, alternating: KGWAS_MX32=0 (16x16x128) against =1 (32x32x64).
# Usage: tools/ab_mx32.sh [bench.py arguments]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab32
for i in 1 2 3; do
  for v in 0 1; do
    KGWAS_MX32=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-subrecords "$@" > gpurun_out/ab32/v$v.$i.json 2>/dev/null
    python3 - $v $i <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/ab32/v%s.%s.json' % (sys.argv[1], sys.argv[2])) if l.startswith('{')][-1])
r = d['roofline']
print('mx32=%s run %s: step %.2f ms | filter %.3f ms/step (frac %.3f) | all kernels %.2f' % (sys.argv[1], sys.argv[2], d['ms_per_step'], r['kernel_ms_per_step'], r['frac'], r['all_scoring_kernels_ms_per_step']))
PY
  done
done
