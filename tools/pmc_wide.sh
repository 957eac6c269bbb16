#!/bin/bash
# SQ counters of the wide kernel's steady launches at the configs[3] shape (separate passes, kernel-trace only).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_wide; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_BUSY_CU_CYCLES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -f csv -d $O/p$i -- python bench.py --samples 2048 --perms 200 --rows 40000000 --steps 1 --warmup 0 --no-cpu-baseline --no-subrecords > $O/p$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for kn in ("wide_kernel", "coarse_kernel<4, 1>"):
    print(kn)
    for f in sorted(glob.glob('gpurun_out/pmc_wide/p*/*/*counter_collection.csv')):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kn in r['Kernel_Name'] and int(r['Grid_Size']) >= 2000000:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items(): print("  %-32s %16.0f  (%d launches)" % (k, sum(v)/len(v), len(v)))
PY
