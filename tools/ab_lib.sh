#!/bin/bash
# A/B of two builds of libkgwas.so on ONE box, alternating (boxes differ by more than most changes): KGWAS_LIB=gpurun_ab/libkgwas_old.so
# against the tree's library. Usage: tools/ab_lib.sh [bench.py arguments]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab
for i in 1 2 3; do
  for v in old new; do
    if [ $v = old ]; then export KGWAS_LIB=$PWD/gpurun_ab/libkgwas_old.so; else unset KGWAS_LIB; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-subrecords "$@" > gpurun_out/ab/$v$i.json 2>/dev/null
    python3 - $v$i <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/ab/%s.json' % sys.argv[1]) if l.startswith('{')][-1])
h = d['host']
print(sys.argv[1], 'step %.2f median %.2f max %.2f | busiest %.2f mean %.2f | dense %.2f tail %.2f | kernels %.2f' % (d['ms_per_step'], h['step_ms_median'], h['step_ms_max'],
      h['replay_worker_busy_ms']['max'], h['replay_worker_busy_ms']['mean'], h['dense_phase_ms_per_step'], h['replay_tail_ms_per_step'], d['roofline']['all_scoring_kernels_ms_per_step']))
PY
  done
done
