#!/bin/bash
# placement / thread-count experiment for the replay pool: ms/step, replay ms, cgroup throttling during the timed region
run() {
  KGWAS_PIN_THREADS=$1 KGWAS_HOST_THREADS=$2 timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['host']; print('pin $1 threads $2: %.1f ms/step, replay %.1f, steps %s throttled %d x %.1f ms' % (d['ms_per_step'], h['replay_ms_per_step'], h['step_ms'], h['cgroup_nr_throttled'], h['cgroup_throttled_ms']))"
}
for round in 1 2 3; do
  run 1 16; run 1 15; run 1 14; run 1 12; run 0 16
done
