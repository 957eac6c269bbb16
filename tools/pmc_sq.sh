#!/bin/bash
# SQ counter passes over one kernel (separate runs, kernel-trace only).
# usage: pmc_sq.sh "<kernel substring>" <grid size of the launches to average> "<bench args>" "<counters>" ["<counters>" ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K="$1"; G="$2"; BARGS="$3"; shift 3
O=gpurun_out/pmc_sq; rm -rf $O; mkdir -p $O
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -f csv -d $O/p$i -- python bench.py $BARGS --steps 1 --warmup 0 --no-cpu-baseline > $O/p$i.log 2>&1
done
K="$K" G="$G" python - <<'PY'
import csv, glob, collections, os
for f in sorted(glob.glob('gpurun_out/pmc_sq/p*/*/*counter_collection.csv')):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if os.environ['K'] in r['Kernel_Name'] and int(r['Grid_Size'])==int(os.environ['G']):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(k, sum(v)/len(v), len(v))
PY
