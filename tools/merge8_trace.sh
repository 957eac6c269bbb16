#!/bin/bash
# Eight ranks on ONE GPU (gloo) at the configs[3] shape with few rows per rank: the breakdown of merge_by_column
# (KGWAS_TRACE) - the part of the 8-GPU step that no single-GPU run shows. The exchange itself runs over gloo here.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/merge8; mkdir -p $O
KGWAS_TRACE_MERGE=1 KGWAS_DIST_BACKEND=gloo timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node ${RANKS:-8} --master-addr 127.0.0.1 --master-port 29544 \
  bench.py --gpus ${RANKS:-8} --rows ${ROWS:-12000000} --steps 2 --warmup 1 > $O/line.json 2> $O/err.txt
grep "merge_by_column\|merge_to_root" $O/err.txt | tail -24
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/merge8/line.json') if l.startswith('{')][-1])
print(d['ms_per_step'], d['n_gpus'], d.get('parity_check'))
for r in d['ranks']: print({k:(round(v,1) if isinstance(v,float) else v) for k,v in r.items() if k in ('rank','step_ms','kernels_ms','replay_ms','merge_ms','replay_threads')})
PY
