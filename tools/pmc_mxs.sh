#!/bin/bash
# SQ counters of the steady filter launches at 2048 x 201: the streaming kernel against the resident one.
cd "$GRAFT_REPO_ROOT"
B="--samples 2048 --perms 200 --rows 40000000 --no-subrecords"
S1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD"
S2="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT"
S3="SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"
KGWAS_MXS=1 bash tools/pmc_mx.sh mxs_kernel 1048576 "$B" "$S1" "$S2" "$S3" > gpurun_out/pmc_mxs_stream.txt 2>&1
cp gpurun_out/pmc_mx/summary.txt gpurun_out/pmc_mxs_stream_summary.txt
KGWAS_MXS=0 bash tools/pmc_mx.sh mx_kernel 7864320 "$B" "$S1" "$S2" "$S3" > gpurun_out/pmc_mxs_resident.txt 2>&1
cp gpurun_out/pmc_mx/summary.txt gpurun_out/pmc_mxs_resident_summary.txt
echo STREAM; cat gpurun_out/pmc_mxs_stream_summary.txt; echo RESIDENT; cat gpurun_out/pmc_mxs_resident_summary.txt
rocprofv3 -L 2>/dev/null | grep -o "SQ_LDS[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_WAIT[A-Z_]*\|TA_[A-Z_]*BUSY[A-Z_]*\|TCP_[A-Z_]*BUSY[A-Z_]*\|SQ_INSTS_VMEM[A-Z_]*\|SQ_INSTS_MFMA[A-Z_]*\|SQ_VALU_MFMA[A-Z_]*" | sort -u | tr '\n' ' '
