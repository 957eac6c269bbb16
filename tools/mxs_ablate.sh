#!/bin/bash
# Time the variants built by tools/mxs_variants.sh on one box: usage tools/mxs_ablate.sh "<bench args>" name1 name2 ...
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/mxs_abl; mkdir -p $O
ARGS=$1; shift
for rep in 1 2; do
for name in "$@"; do
  if [ $name = tree ]; then unset KGWAS_LIB; else export KGWAS_LIB=$PWD/tools/bin/libkgwas_$name.so; fi
  python bench.py $ARGS --steps 3 --warmup 1 --no-cpu-baseline --no-subrecords > $O/$name$rep.json 2> $O/$name$rep.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name$rep.json") if l.startswith('{')][-1]); r=d["roofline"]
    print("%-14s rep $rep step %.2f filter %.2f all_kernels %.2f" % ("$name", d["ms_per_step"], r["kernel_ms_per_step"], r["all_scoring_kernels_ms_per_step"]))
except Exception as e:
    print("$name rep $rep failed:", e, open("$O/$name$rep.err").read()[-300:])
PY
done; done
