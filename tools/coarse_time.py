"""Coarse-filter launch times of one pass over the bench workload (100 M x 1024 x 101, top-10001) for the library
named by KGWAS_LIB (tools/coarse_variants.sh). Ablated variants give wrong results; only the times are read."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmersgwas_amd as kg
from bench import make_phenotypes

S, P, M = 1024, 101, 100_000_000
W = 1 + S // 64
Y = make_phenotypes(S, P - 1, 7)
mac = kg.min_count(S, 0.05, 5)
table = torch.empty(M * W, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
kg.synth_rows_device(table.data_ptr(), 0, M, S, 20240601, stream)
torch.cuda.synchronize()
scan = kg.AssociationScan(S, np.arange(S, dtype=np.uint64), Y, 10001, mac, device=0)
for it in range(3):
    scan.reset()
    t0 = time.perf_counter()
    scan.feed_device(table.data_ptr(), M, 0, stream)
    dt = (time.perf_counter() - t0) * 1e3
    st = scan.stats()
print("%s: step %.1f ms | coarse one-slice %d launches %.2f ms (%.0f M rows), two-slice %d launches %.2f ms (%.0f M rows) | "
      "rescore+sort etc %.2f ms | replay %.1f ms" %
      (os.path.basename(os.environ.get("KGWAS_LIB", "default")), dt, st["coarse_mode_launches"][0], st["coarse_mode_ms"][0],
       st["coarse_mode_rows"][0] / 1e6, st["coarse_mode_launches"][1], st["coarse_mode_ms"][1], st["coarse_mode_rows"][1] / 1e6,
       st["score_kernel_ms"] - st["coarse_kernel_ms"], st["replay_ms"]))
