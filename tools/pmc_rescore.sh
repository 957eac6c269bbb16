#!/bin/bash
# SQ counters of rescore_kernel for the library variants named on the command line (tools/bin/libkgwas_<name>.so).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in "$@"; do
  O=gpurun_out/pmc_rescore_$n; rm -rf $O; mkdir -p $O
  KGWAS_LIB=$PWD/tools/bin/libkgwas_$n.so rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -f csv -d $O -- python bench.py --rows 40000000 --steps 1 --warmup 0 --no-cpu-baseline --no-subrecords > $O/log.txt 2>&1
  python3 - $O $n <<'PY'
import csv, glob, collections, sys
f=glob.glob(sys.argv[1]+'/*/*_counter_collection.csv')[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'rescore_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[2], {k: "%.3e" % (sum(sorted(v)[-6:])/6) for k,v in sorted(acc.items())})
f2=glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')[0]
d=sorted(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in csv.DictReader(open(f2)) if 'rescore_kernel' in r['Kernel_Name'])
print(sys.argv[2], "largest durations us", [round(x/1e3) for x in d[-6:]], "sum ms", sum(d)/1e6)
PY
done
