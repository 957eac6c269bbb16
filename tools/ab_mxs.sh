#!/bin/bash
# A/B of the block-scaled filter's forms on one box: resident LDS groups (KGWAS_MXS=0) against the operand-streaming kernel
# (KGWAS_MXS=1; KGWAS_MXS_FORM=0 two column groups of 7 tiles per block, =1 13 tiles x 32 rows per wave, =2 one wave per SIMD). usage: tools/ab_mxs.sh [samples perms rows]
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/ab_mxs; mkdir -p $O
S=${1:-2048}; P=${2:-200}; R=${3:-100000000}
for rep in 1 2; do
for v in "0 0" "2 0" "2 1" "2 2"; do
  set -- $v
  KGWAS_MXS=$1 KGWAS_MXS_FORM=$2 python bench.py --samples $S --perms $P --rows $R --steps 3 --warmup 1 --no-cpu-baseline --no-subrecords > $O/line_${S}_$1_$2_$rep.json 2> $O/err_${S}_$1_$2_$rep.txt
  python - <<PY
import json
d=json.loads(open("$O/line_${S}_$1_$2_$rep.json").read().strip().splitlines()[-1])
r=d.get("roofline",{})
print("S=$S MXS=$1 FORM=$2 rep=$rep ms_per_step=%.2f filter_ms=%.2f all_kernels=%.2f frac=%.3f parity=%s" % (d["ms_per_step"], r.get("kernel_ms_per_step",-1), r.get("all_scoring_kernels_ms_per_step",-1), r.get("frac",-1), d.get("parity_check")))
PY
done; done
