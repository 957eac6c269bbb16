#!/bin/bash
# the per-GPU workload of the 8-GPU run (250 M rows x 2048 x 201) and the headline with 2 replay threads
B="--no-cpu-baseline --no-subrecords"
for tc in auto 0 1; do
if [ $tc = auto ]; then unset KGWAS_TIE_CHECKS; else export KGWAS_TIE_CHECKS=$tc; fi
echo "== KGWAS_TIE_CHECKS=$tc"
for t in 2 16; do
KGWAS_HOST_THREADS=$t python bench.py --samples 2048 --perms 200 --rows 250000000 --steps 3 --warmup 1 $B | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host']; r=d['roofline']; print('shard threads $t:', round(d['ms_per_step'],2), 'kernels', round(r['all_scoring_kernels_ms_per_step'],1), 'busiest', round(h['replay_ms_per_step'],1), 'cpu', round(h['replay_cpu_ms_per_step'],1), 'tail', round(h['replay_tail_ms_per_step'],2))"
done
KGWAS_HOST_THREADS=2 python bench.py --steps 5 --warmup 2 $B | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host']; print('headline threads 2:', round(d['ms_per_step'],2), 'busiest', round(h['replay_ms_per_step'],1), 'cpu', round(h['replay_cpu_ms_per_step'],1))"
python bench.py --steps 10 --warmup 3 $B | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host']; print('headline threads 16:', round(d['ms_per_step'],2), 'busiest', round(h['replay_ms_per_step'],1), 'cpu', round(h['replay_cpu_ms_per_step'],1))"
done
