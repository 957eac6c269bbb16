#!/bin/bash
# SQ counter passes over kin_gram_kernel (separate runs, kernel-trace only): tools/pmc_kin.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_kin; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CU_CYCLES" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  KIN_CPU_ROWS=100 KGWAS_KIN_NO_TR=${NO_TR:-} rocprofv3 --kernel-trace --pmc $set -f csv -d $O/p$i -- python tools/kin_line.py > $O/p$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_kin/p*/*/*counter_collection.csv')):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'kin_gram' in r['Kernel_Name'] and int(r['Grid_Size']) >= 100000:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print("%-32s %16.0f  (%d launches)" % (k, sum(v)/len(v), len(v)))
PY
