"""bench.py's default_topn sub-record alone (the reference's default -n 1 000 000 on the headline table x 5 columns)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kmersgwas_amd as kg
from bench import make_phenotypes, default_topn_record, usable_cpus
S, M = 1024, 100_000_000
W = 1 + S // 64
Y = make_phenotypes(S, 100, 7)
mac = kg.min_count(S, 0.05, 5)
table = torch.empty(M * W, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
kg.synth_rows_device(table.data_ptr(), 0, M, S, 20240601, stream)
torch.cuda.synchronize()
r = default_topn_record(kg, torch, table, stream, M, S, Y, mac, 0, usable_cpus())
print(json.dumps({k: r[k] for k in ("ms_per_100M_rows", "replay_busiest_worker_ms", "replay_cpu_ms_per_step", "heap_pushes_per_step", "all_scoring_kernels_ms_per_step", "dense_phase_ms_per_step")}))
