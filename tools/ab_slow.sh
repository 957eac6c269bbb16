#!/bin/bash
# One replay worker slowed down (KGWAS_DEBUG_SLOW_WORKER=w:pct, a stand-in for a co-tenant on its pinned CPU) against the
# policies for column groups that fall behind: none / cut up at the tail once other workers have run out of work / also floated whole in mid-scan.
cd "$GRAFT_REPO_ROOT"
for slow in "" "KGWAS_DEBUG_SLOW_WORKER=5:30" "KGWAS_DEBUG_SLOW_WORKER=5:100"; do
for pol in "KGWAS_SPLIT_LAGGING=0 KGWAS_FLOAT_LEAD=0" "KGWAS_SPLIT_LAGGING=1 KGWAS_FLOAT_LEAD=0" "KGWAS_SPLIT_LAGGING=1 KGWAS_FLOAT_LEAD=2"; do
  v="$slow $pol"
  env $v python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-subrecords $AB_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); h=d['host']
print('%-60s step %.2f median %.2f max %.2f | busiest %.2f mean %.2f cpu %.1f splits %.2f' % ('$v', d['ms_per_step'], h['step_ms_median'], h['step_ms_max'], h['replay_worker_busy_ms']['max'], h['replay_worker_busy_ms']['mean'], h['replay_cpu_ms_per_step'], h['replay_group_splits_per_step']))"
done; done
