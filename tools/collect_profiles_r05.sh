#!/bin/bash
# Collect the rocprofv3 evidence of round 5 on the GPU box (outputs under gpurun_out/prof_r05/).
# Kernel timing and PMC counters are collected in separate runs (never --pmc together with API/sys traces).
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r05
rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-subrecords"
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
SQ2="SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
# 1. the bench workload (configs[1]): kernel stats, PMC HBM traffic of the block-scaled filter (mx_kernel)
rocprofv3 --kernel-trace --stats -f csv -d $O/stats -- python bench.py --steps 3 --warmup 1 $B > $O/stats_bench.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc_fetch -- python bench.py --rows 40000000 --steps 1 --warmup 0 $B > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/pmc_write -- python bench.py --rows 40000000 --steps 1 --warmup 0 $B > $O/pmc_write.log 2>&1
python tools/publish_profiles_r05.py traffic
# 2. the bench line itself (all sub-records; first among the rest: if the call runs out of time, the judged record exists)
python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err
tail -c 300 $O/bench_line.json
# 3. SQ counters: the resident filter at 1024 x 101, the streaming filter at 2048 x 201
rocprofv3 --kernel-trace --pmc $SQ1 -f csv -d $O/pmc_sq1 -- python bench.py --rows 40000000 --steps 1 --warmup 0 $B > $O/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ2 -f csv -d $O/pmc_sq2 -- python bench.py --rows 40000000 --steps 1 --warmup 0 $B > $O/pmc_sq2.log 2>&1
KGWAS_MXS=2 rocprofv3 --kernel-trace --pmc $SQ1 -f csv -d $O/pmc_c3_sq1 -- python bench.py --samples 2048 --perms 200 --rows 40000000 --steps 1 --warmup 0 $B > $O/pmc_c3_sq1.log 2>&1
KGWAS_MXS=2 rocprofv3 --kernel-trace --pmc $SQ2 -f csv -d $O/pmc_c3_sq2 -- python bench.py --samples 2048 --perms 200 --rows 40000000 --steps 1 --warmup 0 $B > $O/pmc_c3_sq2.log 2>&1
# 4. the one-column scan (narrow filter): kernel stats at 100 M rows; at 1.2 G rows (163 GB): kernel stats + PMC traffic
rocprofv3 --kernel-trace --stats -f csv -d $O/p1_stats -- python tools/one_column.py > $O/p1_stats.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $O/p1l_stats -- python tools/p1_large_once.py 1200000000 2 > $O/p1l_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/p1l_fetch -- python tools/p1_large_once.py 1200000000 1 > $O/p1l_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/p1l_write -- python tools/p1_large_once.py 1200000000 1 > $O/p1l_write.log 2>&1
# 5. kinship
KIN_CPU_ROWS=500 rocprofv3 --kernel-trace --stats -f csv -d $O/kin_stats -- python tools/kin_line.py > $O/kin_stats.log 2>&1
# 6. configs[3] shape on one GPU (100 M rows): the default (resident plan: five LDS groups) and the streaming filter (KGWAS_MXS=2):
#    lines, kernel stats, HBM-side traffic of both
python bench.py --samples 2048 --perms 200 --rows 100000000 --steps 5 --warmup 2 $B > $O/config4_line.json 2> $O/config4_line.err
KGWAS_MXS=2 python bench.py --samples 2048 --perms 200 --rows 100000000 --steps 5 --warmup 2 $B > $O/config4_streaming_line.json 2> $O/config4_streaming_line.err
rocprofv3 --kernel-trace --stats -f csv -d $O/c3_stats -- python bench.py --samples 2048 --perms 200 --rows 100000000 --steps 2 --warmup 1 $B > $O/c3_stats.log 2>&1
KGWAS_MXS=2 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc_c3_fetch -- python bench.py --samples 2048 --perms 200 --rows 40000000 --steps 1 --warmup 0 $B > $O/pmc_c3_fetch.log 2>&1
KGWAS_MXS=2 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/pmc_c3_write -- python bench.py --samples 2048 --perms 200 --rows 40000000 --steps 1 --warmup 0 $B > $O/pmc_c3_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc_c3r_fetch -- python bench.py --samples 2048 --perms 200 --rows 40000000 --steps 1 --warmup 0 $B > $O/pmc_c3r_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/pmc_c3r_write -- python bench.py --samples 2048 --perms 200 --rows 40000000 --steps 1 --warmup 0 $B > $O/pmc_c3r_write.log 2>&1
# 7. the per-GPU workload of the 8-GPU run on one GPU (250 M rows x 2048 x 201): 16 replay threads, and the 2 a rank gets under a 16-CPU quota
python bench.py --samples 2048 --perms 200 --rows 250000000 --steps 5 --warmup 2 $B > $O/shard250M_line.json 2> $O/shard250M_line.err
KGWAS_HOST_THREADS=2 python bench.py --samples 2048 --perms 200 --rows 250000000 --steps 3 --warmup 1 $B > $O/shard250M_2threads_line.json 2> $O/shard250M_2threads_line.err
find $O -name "*.csv" | xargs ls -la | awk '{print $5, $9}' | tail -40
python tools/publish_profiles_r05.py
