#!/bin/bash
# SQ counter passes over the steady launches of one kernel (separate runs, kernel-trace only) + their durations.
# usage: pmc_mx.sh "<kernel substring>" <grid size of the launches to average> "<bench args>" "<counters>" ["<counters>" ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K="$1"; G="$2"; BARGS="$3"; shift 3
O=gpurun_out/pmc_mx; rm -rf $O; mkdir -p $O
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -f csv -d $O/p$i -- python bench.py $BARGS --steps 1 --warmup 0 --no-cpu-baseline > $O/p$i.log 2>&1
done
K="$K" G="$G" python - <<'PY' | tee gpurun_out/pmc_mx/summary.txt
import csv, glob, collections, os
K, G = os.environ['K'], int(os.environ['G'])
for f in sorted(glob.glob('gpurun_out/pmc_mx/p*/*/*counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if K in r['Kernel_Name'] and int(r['Grid_Size']) == G:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()): print("%-36s %16.0f  (%d launches)" % (k, sum(v) / len(v), len(v)))
for f in sorted(glob.glob('gpurun_out/pmc_mx/p*/*/*kernel_trace.csv')):
    d = [ (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in csv.DictReader(open(f)) if K in r['Kernel_Name'] and int(r.get('Grid_Size', r.get('Grid_Size_X'))) == G ]
    if d: print("%s: %d launches, avg duration %.1f us" % (f.split('/')[2], len(d), sum(d) / len(d) / 1e3))
PY
