#!/bin/bash
# Do the record copies (hipMemcpyAsync device -> pinned ring) run as shader blits (__amd_rocclr_copyBuffer, beside the scan's kernels on
# the CUs) or on the SDMA engines? Registered huge-page memory (default) against hipHostMalloc (KGWAS_PIN_PLAIN=1): kernel trace of one
# step each, then the headline step alternating.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/abpin
for n in reg plain; do
  if [ $n = plain ]; then export KGWAS_PIN_PLAIN=1; else unset KGWAS_PIN_PLAIN; fi
  rm -rf gpurun_out/abpin/kt_$n
  rocprofv3 --kernel-trace -f csv -d gpurun_out/abpin/kt_$n -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-subrecords > /dev/null 2>&1
  python tools/chunk_timeline.py gpurun_out/abpin/kt_$n > gpurun_out/abpin/timeline_$n.txt
  echo "== $n: copyBuffer launches in the traced step: $(grep -c copyBuffer gpurun_out/abpin/timeline_$n.txt)"; tail -1 gpurun_out/abpin/timeline_$n.txt
done
: > gpurun_out/abpin/times.txt
for i in 1 2 3; do
  for n in reg plain; do
    if [ $n = plain ]; then export KGWAS_PIN_PLAIN=1; else unset KGWAS_PIN_PLAIN; fi
    echo "== $n $i" >> gpurun_out/abpin/times.txt
    timeout 300 python bench.py --no-cpu-baseline --no-subrecords --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('bench: ms_per_step %.2f median %.2f | mx %.2f all kernels %.2f' % (d['ms_per_step'], d.get('ms_per_step_median', 0), r['kernel_ms_per_step'], r['all_scoring_kernels_ms_per_step']))" >> gpurun_out/abpin/times.txt
  done
done
cat gpurun_out/abpin/times.txt
