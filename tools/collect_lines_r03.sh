#!/bin/bash
# The two bench lines of profiles/ alone (no profiler): python bench.py as the driver runs it, and the configs[3] shape on one GPU.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/prof_r03; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err
python bench.py --samples 2048 --perms 200 --rows 100000000 --steps 5 --warmup 2 --no-cpu-baseline --no-subrecords > $O/config4_line.json 2> $O/config4_line.err
python3 - <<'PY'
import json
for n in ('bench_line', 'config4_line'):
    d = json.loads([l for l in open('gpurun_out/prof_r03/%s.json' % n) if l.startswith('{')][-1])
    print(n, round(d['ms_per_step'], 2), d['host']['step_ms_median'], d['host']['replay_worker_busy_ms'])
PY
