#!/bin/bash
# kernel + memory-copy timeline of the last one-column pass of tools/p1_trace.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/p1prof; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace -f csv -d $O -- python tools/p1_trace.py ${1:-1} > $O/run.log 2>&1
python - <<'PY'
import csv, glob
ev = []
for f in glob.glob('gpurun_out/p1prof/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'][:60]))
for f in glob.glob('gpurun_out/p1prof/*/*memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'M %s %s bytes' % (r.get('Direction', ''), r.get('Size', r.get('Bytes', '?')))))
ev.sort()
# the last pass: everything after the last dense-select kernel
idx = max(i for i, e in enumerate(ev) if 'dense_select' in e[2] or 'score_mfma' in e[2])
while idx > 0 and ev[idx][0] - ev[idx - 1][1] < 200000: idx -= 1
t0 = ev[idx][0]
with open('gpurun_out/p1prof/timeline.txt', 'w') as out:
    for s, e, n in ev[idx:]:
        out.write("%9.3f %9.3f  %7.1f us  %s\n" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3, n))
print(open('gpurun_out/p1prof/timeline.txt').read()[:12000])
PY
