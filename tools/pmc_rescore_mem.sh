#!/bin/bash
# Memory-side counters of rescore_kernel's largest launches (separate passes, kernel-trace only; every pass under its own timeout:
# the TA_* counter set hung the profiler for the whole call in round 6 and was taken out).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_rescore_mem; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_SALU" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set -f csv -d $O/p$i -- python bench.py --rows 40000000 --steps 1 --warmup 0 --no-cpu-baseline --no-subrecords > $O/p$i.log 2>&1
  tail -2 $O/p$i.log | cut -c1-200
done
python3 - <<'PY'
import csv, glob, collections
for p in sorted(glob.glob('gpurun_out/pmc_rescore_mem/p*/')):
    cf = glob.glob(p + '*/*_counter_collection.csv'); kf = glob.glob(p + '*/*_kernel_trace.csv')
    if not cf or not kf: print(p, "no output"); continue
    dur = {int(r['Dispatch_Id']): int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(kf[0])) if 'rescore_kernel' in r['Kernel_Name']}
    top = sorted(dur, key=dur.get)[-4:]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(cf[0])):
        if int(r['Dispatch_Id']) in top: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(p, "4 longest rescore launches: avg %.0f us" % (sum(dur[t] for t in top) / 4e3))
    for k, v in sorted(acc.items()): print("   %-36s %14.0f" % (k, sum(v) / len(v)))
PY
