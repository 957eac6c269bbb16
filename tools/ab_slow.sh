#!/bin/bash
# One replay worker slowed down (KGWAS_DEBUG_SLOW_WORKER=w:pct, a stand-in for a co-tenant on its pinned CPU), with and
# without the splitting of lagging column groups; and the undisturbed scan both ways.
cd "$GRAFT_REPO_ROOT"
for v in "KGWAS_SPLIT_LAGGING=1" "KGWAS_SPLIT_LAGGING=0" "KGWAS_DEBUG_SLOW_WORKER=5:100 KGWAS_SPLIT_LAGGING=1" "KGWAS_DEBUG_SLOW_WORKER=5:100 KGWAS_SPLIT_LAGGING=0" \
         "KGWAS_DEBUG_SLOW_WORKER=5:30 KGWAS_SPLIT_LAGGING=1" "KGWAS_DEBUG_SLOW_WORKER=5:30 KGWAS_SPLIT_LAGGING=0" "KGWAS_SPLIT_LAGGING=1" "KGWAS_SPLIT_LAGGING=0"; do
  env $v python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-subrecords $AB_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); h=d['host']
print('%-60s step %.2f median %.2f max %.2f | busiest %.2f mean %.2f | parity %s' % ('$v', d['ms_per_step'], h['step_ms_median'], h['step_ms_max'], h['replay_worker_busy_ms']['max'], h['replay_worker_busy_ms']['mean'], d.get('parity_check')))"
done
