#!/bin/bash
# Collect the rocprofv3 evidence for the bench workload on the GPU box (outputs under gpurun_out/prof_r01/).
# Kernel timing and PMC counters are collected in separate runs (never --pmc together with API/sys traces).
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r01
mkdir -p $O
rocprofv3 --kernel-trace --stats -f csv -d $O/stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/stats_bench.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc_fetch -- python bench.py --rows 40000000 --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/pmc_write -- python bench.py --rows 40000000 --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_write.log 2>&1
python bench.py > $O/bench_line.json 2> $O/bench_line.err
tail -c 600 $O/bench_line.json
find $O -name "*.csv" | xargs ls -la
