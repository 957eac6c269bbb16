#!/bin/bash
# Collect the rocprofv3 evidence of round 6 on the GPU box (outputs under gpurun_out/prof_r06/).
# Kernel timing and PMC counters are collected in separate runs (never --pmc together with API/sys traces).
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r06
rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-subrecords"
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
SQ2="SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
# 0. what the matrix pipe sustains for seconds, board power and clock beside it; the same readings around the headline feed
timeout 120 tools/bin/probe_mx_power 3 > $O/probe_mx_power.txt 2>&1
timeout 200 python tools/power_trace.py mx 6 2>/dev/null | tail -1 > $O/power_trace_mx.json
# 1. the bench workload (configs[1]): kernel stats, PMC HBM traffic and SQ counters of the block-scaled filter (mx_kernel, persistent blocks)
rocprofv3 --kernel-trace --stats -f csv -d $O/stats -- python bench.py --steps 3 --warmup 1 $B > $O/stats_bench.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc_fetch -- python bench.py --rows 40000000 --steps 1 --warmup 0 $B > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/pmc_write -- python bench.py --rows 40000000 --steps 1 --warmup 0 $B > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ1 -f csv -d $O/pmc_sq1 -- python bench.py --rows 40000000 --steps 1 --warmup 0 $B > $O/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ2 -f csv -d $O/pmc_sq2 -- python bench.py --rows 40000000 --steps 1 --warmup 0 $B > $O/pmc_sq2.log 2>&1
# 2. the bench line itself, every field
python bench.py --steps 20 --warmup 5 --full > $O/bench_line.json 2> $O/bench_line.err
tail -c 300 $O/bench_line.json
# 3. one-column scan and kinship: kernel stats
rocprofv3 --kernel-trace --stats -f csv -d $O/p1_stats -- python tools/one_column.py > $O/p1_stats.log 2>&1
KIN_CPU_ROWS=500 rocprofv3 --kernel-trace --stats -f csv -d $O/kin_stats -- python tools/kin_line.py > $O/kin_stats.log 2>&1
# 4. the north-star shape (2048 x 201) on one GPU: kernel stats at 100 M rows, HBM-side traffic of the default (resident) plan
rocprofv3 --kernel-trace --stats -f csv -d $O/c3_stats -- python bench.py --samples 2048 --perms 200 --rows 100000000 --steps 2 --warmup 1 $B > $O/c3_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc_c3r_fetch -- python bench.py --samples 2048 --perms 200 --rows 40000000 --steps 1 --warmup 0 $B > $O/pmc_c3r_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/pmc_c3r_write -- python bench.py --samples 2048 --perms 200 --rows 40000000 --steps 1 --warmup 0 $B > $O/pmc_c3r_write.log 2>&1
# 5. what the round's second half measured: event packets, the record copies' path, the chunk timeline, steps and one-column passes of the final library
timeout 60 tools/bin/probe_events > $O/probe_events.txt 2>&1
timeout 120 tools/probe_d2h.sh 2>&1 | cut -c1-260 | grep -v "amdgpu.ids" > $O/probe_d2h.txt
rocprofv3 --kernel-trace -f csv -d $O/ktrace -- python bench.py --steps 1 --warmup 1 $B > /dev/null 2>&1
python tools/chunk_timeline.py $O/ktrace > $O/chunk_timeline.txt 2>&1
(python tools/steps_stat.py 200 final; python tools/one_column_passes.py 100) 2>&1 | grep -E "step median|one column" > $O/steps_final.txt
find $O -name "*.csv" | xargs ls -la | awk '{print $5, $9}' | tail -30
python tools/publish_profiles_r06.py
