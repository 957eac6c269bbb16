#!/bin/bash
# A/B on one box, alternating: tools/bin/libkgwas_old.so (events recorded behind the launches) against the tree's library (events bound to
# the launches, launch.h): the headline step and the one-column pass.
mkdir -p gpurun_out/ab
: > gpurun_out/ab/events.txt
for i in 1 2 3; do
  for n in old new; do
    if [ $n = old ]; then export KGWAS_LIB=$PWD/tools/bin/libkgwas_old.so; else unset KGWAS_LIB; fi
    echo "== $n $i" >> gpurun_out/ab/events.txt
    timeout 300 python tools/one_column_passes.py 60 2>&1 | tail -1 >> gpurun_out/ab/events.txt
    timeout 300 python bench.py --no-cpu-baseline --no-subrecords --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('bench: ms_per_step %.2f median %.2f | mx %.2f all kernels %.2f' % (d['ms_per_step'], d.get('ms_per_step_median', 0), r['kernel_ms_per_step'], r['all_scoring_kernels_ms_per_step']))" >> gpurun_out/ab/events.txt
  done
done
cat gpurun_out/ab/events.txt
