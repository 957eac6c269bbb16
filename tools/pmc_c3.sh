#!/bin/bash
# HBM-side read traffic of mx_kernel at the configs[3] shape (2048 x 201: five LDS groups per row block share rows through one XCD's L2)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_c3; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O -- python bench.py --samples 2048 --perms 200 --rows 40000000 --steps 1 --warmup 0 --no-cpu-baseline --no-subrecords > $O/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f=glob.glob('gpurun_out/pmc_c3/*/*_counter_collection.csv')[0]
v=[(float(r['Counter_Value']), int(r['Grid_Size'])) for r in csv.DictReader(open(f)) if 'mx_kernel' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE']
big=[x for x in v if x[1]==max(y[1] for y in v)]
kib=sum(x[0] for x in big)/len(big)
print("launches", len(v), "largest grid", big[0][1], "n", len(big), "FETCH_SIZE KiB avg", kib, "-> HBM-side read bytes", 2*kib*1024)
PY
grep -c . $O/log.txt
