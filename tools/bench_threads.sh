#!/bin/bash
# replay-pool size experiment: the cgroup quota is CPU TIME (16 CPUs here) while 256 logical CPUs are schedulable, so
# more than 16 pinned workers shorten each chunk's replay (fewer columns per worker) as long as the average stays
# under the quota; watch the throttling counters.
for round in 1 2 3; do
for t in 16 17 18 20 21 26; do
  KGWAS_HOST_THREADS=$t timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['host']; print('threads $t round $round: %.1f ms/step, replay %.1f, steps %s throttled %d x %.1f ms' % (d['ms_per_step'], h['replay_ms_per_step'], h['step_ms'], h['cgroup_nr_throttled'], h['cgroup_throttled_ms']))"
done; done
