"""Breakdown of a one-column scan (100 M x 1024, top-10001): the HBM-bound side of the coarse filter."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kmersgwas_amd as kg
from bench import make_phenotypes
S, P, M = 1024, 1, 100_000_000
W = 1 + S // 64
Y = make_phenotypes(S, P - 1, 7) if P > 1 else make_phenotypes(S, 0, 7)
mac = kg.min_count(S, 0.05, 5)
table = torch.empty(M * W, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
kg.synth_rows_device(table.data_ptr(), 0, M, S, 20240601, stream)
torch.cuda.synchronize()
scan = kg.AssociationScan(S, np.arange(S, dtype=np.uint64), Y[:P], 10001, mac, device=0)
for it in range(4):
    scan.reset()
    t0 = time.perf_counter()
    scan.feed_device(table.data_ptr(), M, 0, stream)
    t1 = time.perf_counter()
    scan.finish()
    t2 = time.perf_counter()
    st = scan.stats()
print("feed %.2f ms finish %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
for k in ("score_kernel_ms", "coarse_kernel_ms", "coarse_launches", "coarse_mode_ms", "coarse_mode_launches", "coarse_mode_rows", "coarse_mode_tiles", "replay_ms", "gpu_wait_ms", "dense_ms", "chunks", "candidates", "heap_pushes"):
    print(k, st[k])
