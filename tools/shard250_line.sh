#!/bin/bash
# The per-GPU workload of the 8-GPU configuration (BASELINE.json configs[3]: 250 M rows x 2048 samples x 201 columns per
# GPU) on ONE GPU, and the same shape through the N > 1 path with two ranks sharing the GPU (gloo; 60 M rows each).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/shard250; mkdir -p $O
python bench.py --samples 2048 --perms 200 --rows 250000000 --steps 3 --warmup 1 --no-cpu-baseline --no-subrecords > $O/line.json 2> $O/err.txt
tail -c 400 $O/line.json; echo
KGWAS_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --rows 60000000 --steps 2 --warmup 1 > $O/line_2ranks.json 2> $O/err_2ranks.txt
tail -c 600 $O/line_2ranks.json; echo; tail -3 $O/err_2ranks.txt
