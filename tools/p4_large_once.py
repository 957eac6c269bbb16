"""Four-column scan over a large resident table (narrow filter, all four column slots in use): pass and filter times."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kmersgwas_amd as kg
from bench import make_phenotypes
S, P = 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 600_000_000
W = 1 + S // 64
Y = make_phenotypes(S, P - 1, 7)
mac = kg.min_count(S, 0.05, 5)
table = torch.empty(rows * W, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
kg.synth_rows_device(table.data_ptr(), 0, rows, S, 20240601, stream)
torch.cuda.synchronize()
scan = kg.AssociationScan(S, np.arange(S, dtype=np.uint64), Y, 10001, mac, device=0)
ts, fs = [], []
for it in range(4):
    scan.reset(); scan.expect_finish()
    t0 = time.perf_counter()
    scan.feed_device(table.data_ptr(), rows, 0, stream); scan.finish()
    if it:
        ts.append((time.perf_counter() - t0) * 1e3); fs.append(scan.stats()["coarse_kernel_ms"])
b = rows * 8.0 * W
print("P=%d rows=%d: pass min %.2f ms = %.3f of 8 TB/s; filter %.2f ms = %.3f (%.2f TB/s)" % (P, rows, min(ts), b / (min(ts) * 1e-3) / 8e12, min(fs), b / (min(fs) * 1e-3) / 8e12, b / (min(fs) * 1e-3) / 1e12))
