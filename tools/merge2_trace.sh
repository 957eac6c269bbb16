#!/bin/bash
# Two ranks on one GPU (gloo), 8 replay threads each, the column-distributed merge forced: per-phase costs with a
# realistic thread count per rank.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/merge2; mkdir -p $O
KGWAS_BENCH_MERGE=column KGWAS_BENCH_MERGE_DIAG=1 KGWAS_TRACE_MERGE=1 KGWAS_DIST_BACKEND=gloo timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29545 \
  bench.py --gpus 2 --rows ${ROWS:-40000000} --steps 3 --warmup 1 > $O/line.json 2> $O/err.txt
grep "merge_by_column\|merge_to_root" $O/err.txt | tail -8
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/merge2/line.json') if l.startswith('{')][-1])
print(d['ms_per_step'], d['n_gpus'], d.get('parity_check'))
for r in d['ranks']: print({k:(round(v,1) if isinstance(v,float) else v) for k,v in r.items() if k in ('rank','step_ms','kernels_ms','replay_ms','merge_ms','replay_threads')})
PY
