#!/bin/bash
# placement experiment for the replay pool (KGWAS_PIN_THREADS: 0 none, 1 one core each, 2 node share, 3 L3 domain)
for round in 1 2 3 4; do
for pin in 0 1 2 3; do
  KGWAS_PIN_THREADS=$pin timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['host']; print('pin $pin round $round: %.1f ms/step, replay %.1f, steps %s' % (d['ms_per_step'], h['replay_ms_per_step'], h['step_ms']))"
done; done
